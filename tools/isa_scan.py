"""Scan the device assembly of every HIP translation unit for two latency patterns that hide in C++ source (round 5):
  * serialized loads -- a global_load with a full `s_waitcnt vmcnt(0)` (or vmcnt(1)) right behind it, many times in one kernel: a bounds test
    around a load (the compiler gives each its own branch and wait), or a descriptor field read through a reference next to stores (re-fetched
    before every access);
  * table loads -- many loads off one scalar base with small constant offsets: a constexpr table indexed by `cond ? a : b` that became memory.
usage (no GPU needed): python tools/isa_scan.py [min_count=4]        -> kernels with at least that many serialized loads, and their load totals
(spin loops that poll a flag -- the yield table -- show up as a handful of hits per kernel and are expected)."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'hyperbo_amd', 'csrc')
min_count = int(sys.argv[1]) if len(sys.argv) > 1 else 4
vgpr_form = {'chol.hip', 'small.hip'}
for f in sorted(os.listdir(src)):
    if not f.endswith('.hip'):
        continue
    with tempfile.NamedTemporaryFile(suffix='.s') as tmp:
        cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-munsafe-fp-atomics', '--cuda-device-only', '-S']
        if f in vgpr_form:
            cmd += ['-mllvm', '-amdgpu-mfma-vgpr-form']
        if subprocess.run(cmd + [os.path.join(src, f), '-o', tmp.name], stderr=subprocess.DEVNULL).returncode:
            print(f, 'compile failed'); continue
        lines = open(tmp.name).read().splitlines()
    cur, res = None, {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            cur = m.group(1); res[cur] = [0, 0]
        if cur and 'global_load' in l:
            res[cur][1] += 1
            for j in range(i + 1, min(i + 5, len(lines))):
                if 'global_load' in lines[j]:
                    break
                if 's_waitcnt vmcnt(0)' in lines[j] or 's_waitcnt vmcnt(1)' in lines[j]:
                    res[cur][0] += 1; break
    for k, (a, b) in res.items():
        if a >= min_count:
            try:
                name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip() or k
            except OSError:
                name = k
            print('%-14s %-90s serialized %3d of %3d loads' % (f, name[:90], a, b))

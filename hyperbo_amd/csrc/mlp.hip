// The MLP basis of a batch of tasks in one launch per layer (hyperbo/gp_utils/basis_functions.py:24-36: flax Dense + tanh,
// shared weights, applied to every sub-dataset's inputs) and its backward pass (what jax.value_and_grad of gp.py:134 does to it).
// The per-task forms (gram.hip: dense_tanh_kernel, grad.hip: dense_bwd_*) cost 1 + 3 launches per task and layer: a pre-training
// step over 24 sub-datasets of 100 points with a two-layer basis was 190 launches of ~5 us, 1.4 of its 1.6 ms.  Same arithmetic per
// element, in the same order (the weight gradients are summed with fp64 atomics, as before).
#include "hbo_internal.h"

namespace {
// out[i][o] = tanh(sum_k in[i][k] w[k][o] + b[o]) for every task of the batch; grid.y = task
template <typename T>
__global__ void dense_tanh_batch_kernel(const MlpTaskDev* __restrict__ mt, int layer, const T* __restrict__ w, const T* __restrict__ b,
                                        int fin, int fout) {
  const MlpTaskDev& t = mt[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= t.n * fout) return;
  const T* in = static_cast<const T*>(layer ? t.acts[layer - 1] : t.x);
  T* out = static_cast<T*>(t.acts[layer]);
  const int64_t i = idx / fout;
  const int o = (int)(idx % fout);
  T s = b[o];
  for (int k = 0; k < fin; ++k) s += in[i * fin + k] * w[(int64_t)k * fout + o];
  out[idx] = tanh(s);
}
__global__ void zero_dF_batch_kernel(const MlpTaskDev* __restrict__ mt, int flast) {
  const MlpTaskDev& t = mt[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < t.n * flast) t.dF[idx] = 0.0;
}
// dz = dout * (1 - out^2), in place on the current gradient buffer
template <typename T>
__global__ void dense_bwd_dz_batch_kernel(const MlpTaskDev* __restrict__ mt, int layer, int cur_is_dF, int fout) {
  const MlpTaskDev& t = mt[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= t.n * fout) return;
  double* dout = cur_is_dF ? t.dF : t.dtmp;
  const double o = (double)static_cast<const T*>(t.acts[layer])[idx];
  dout[idx] *= (1.0 - o * o);
}
// dW[k][o] += sum_i in[i][k] dz[i][o]; db[o] += sum_i dz[i][o]   (grid.x = k in 0..fin (fin = bias row), grid.y = row chunk, grid.z = task)
template <typename T>
__global__ void dense_bwd_w_batch_kernel(const MlpTaskDev* __restrict__ mt, int layer, int cur_is_dF, int fin, int fout, double* dW, double* db,
                                         int rows_per_block) {
  const MlpTaskDev& t = mt[blockIdx.z];
  const int k = blockIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * rows_per_block;
  if (i0 >= t.n) return;
  int64_t i1 = i0 + rows_per_block; if (i1 > t.n) i1 = t.n;
  const T* in = static_cast<const T*>(layer ? t.acts[layer - 1] : t.x);
  const double* dz = cur_is_dF ? t.dF : t.dtmp;
  for (int o = threadIdx.x; o < fout; o += blockDim.x) {
    double s = 0;
    if (k < fin) { for (int64_t i = i0; i < i1; ++i) s += (double)in[i * fin + k] * dz[i * fout + o]; atomicAdd(&dW[(int64_t)k * fout + o], s); }
    else { for (int64_t i = i0; i < i1; ++i) s += dz[i * fout + o]; atomicAdd(&db[o], s); }
  }
}
// din[i][k] = sum_o dz[i][o] w[k][o]  into the other gradient buffer
template <typename T>
__global__ void dense_bwd_in_batch_kernel(const MlpTaskDev* __restrict__ mt, int cur_is_dF, const T* __restrict__ w, int fin, int fout) {
  const MlpTaskDev& t = mt[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= t.n * fin) return;
  const double* dz = cur_is_dF ? t.dF : t.dtmp;
  double* din = cur_is_dF ? t.dtmp : t.dF;
  const int64_t i = idx / fin; const int k = (int)(idx % fin);
  double s = 0;
  for (int o = 0; o < fout; ++o) s += dz[i * fout + o] * (double)w[(int64_t)k * fout + o];
  din[idx] = s;
}
}  // namespace

void launch_mlp_forward_batch(int dtype, const MlpTaskDev* mt, int ntasks, int64_t max_n, int layer, const void* w, const void* b, int fin,
                              int fout, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return;
  dim3 grid((unsigned)((max_n * fout + 255) / 256), (unsigned)ntasks);
  if (dtype == HBO_F64) hipLaunchKernelGGL((dense_tanh_batch_kernel<double>), grid, dim3(256), 0, st, mt, layer, (const double*)w, (const double*)b, fin, fout);
  else hipLaunchKernelGGL((dense_tanh_batch_kernel<float>), grid, dim3(256), 0, st, mt, layer, (const float*)w, (const float*)b, fin, fout);
}
void launch_mlp_zero_dF_batch(const MlpTaskDev* mt, int ntasks, int64_t max_n, int flast, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return;
  hipLaunchKernelGGL(zero_dF_batch_kernel, dim3((unsigned)((max_n * flast + 255) / 256), (unsigned)ntasks), dim3(256), 0, st, mt, flast);
}
void launch_dense_bwd_batch(int dtype, const MlpTaskDev* mt, int ntasks, int64_t max_n, int layer, int cur_is_dF, const void* w, double* dW,
                            double* db, int fin, int fout, int want_din, hipStream_t st) {
  if (ntasks <= 0 || max_n <= 0) return;
  const int rpb = 256;
  const dim3 gz((unsigned)((max_n * fout + 255) / 256), (unsigned)ntasks);
  const dim3 gw(fin + 1, (unsigned)((max_n + rpb - 1) / rpb), (unsigned)ntasks);
  const dim3 gi((unsigned)((max_n * fin + 255) / 256), (unsigned)ntasks);
  const int thr = fout < 64 ? 64 : (fout > 256 ? 256 : ((fout + 63) / 64) * 64);
  if (dtype == HBO_F64) {
    hipLaunchKernelGGL((dense_bwd_dz_batch_kernel<double>), gz, dim3(256), 0, st, mt, layer, cur_is_dF, fout);
    hipLaunchKernelGGL((dense_bwd_w_batch_kernel<double>), gw, dim3(thr), 0, st, mt, layer, cur_is_dF, fin, fout, dW, db, rpb);
    if (want_din) hipLaunchKernelGGL((dense_bwd_in_batch_kernel<double>), gi, dim3(256), 0, st, mt, cur_is_dF, (const double*)w, fin, fout);
  } else {
    hipLaunchKernelGGL((dense_bwd_dz_batch_kernel<float>), gz, dim3(256), 0, st, mt, layer, cur_is_dF, fout);
    hipLaunchKernelGGL((dense_bwd_w_batch_kernel<float>), gw, dim3(thr), 0, st, mt, layer, cur_is_dF, fin, fout, dW, db, rpb);
    if (want_din) hipLaunchKernelGGL((dense_bwd_in_batch_kernel<float>), gi, dim3(256), 0, st, mt, cur_is_dF, (const float*)w, fin, fout);
  }
}

/* TEST-ONLY stand-in for librccl: the five entry points libhbo's comm.hip binds (ncclGetUniqueId, ncclCommInitRank,
 * ncclAllReduce, ncclCommDestroy, ncclCommAbort) over POSIX shared memory + hipMemcpy, so that TWO ranks on the ONE GPU of the
 * test box (which RCCL refuses: duplicate device in a communicator) go through hbo_comm_init(rank > 0), hbo_objective_sharded's
 * device-buffer all-reduce, its NaN-contribution path and its abort path.  Loaded through $HBO_RCCL_LIB (comm.hip: rccl_open).
 * Not a collective library: fp64 sum only, host-blocking, ranks of one host.  Built by __graft_entry__.build():
 *   gcc -O2 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -o tests/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt
 */
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#define FAKE_MAX_RANKS 8
#define FAKE_MAX_COUNT 65536
#define FAKE_TIMEOUT_S 30.0

typedef struct { char internal[128]; } fakeUniqueId;

typedef struct {
  volatile int arrived;      /* barrier: arrivals of the current generation */
  volatile int generation;
  volatile int aborted;
  volatile int attached;
  volatile int calls;        /* all-reduces completed (rank 0 counts): the tests read it through fakeRcclCalls */
  double slot[FAKE_MAX_RANKS][FAKE_MAX_COUNT];
} fakeShared;

typedef struct {
  fakeShared* sh;
  int rank, nranks;
  char name[128];
} fakeComm;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* sense-reversing barrier on the shared segment; 0 ok, 1 aborted / timed out */
static int fake_barrier(fakeComm* c) {
  fakeShared* s = c->sh;
  const int gen = s->generation;
  if (__sync_add_and_fetch(&s->arrived, 1) == c->nranks) {
    s->arrived = 0;
    __sync_synchronize();
    __sync_add_and_fetch(&s->generation, 1);
    return s->aborted ? 1 : 0;
  }
  const double t0 = now_s();
  while (s->generation == gen) {
    if (s->aborted) return 1;
    if (now_s() - t0 > FAKE_TIMEOUT_S) { s->aborted = 1; return 1; }
    usleep(50);
  }
  return s->aborted ? 1 : 0;
}

int ncclGetUniqueId(fakeUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/hbofake_%d_%ld", (int)getpid(), (long)(now_s() * 1e6));
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, fakeUniqueId id, int rank) {
  if (!comm || nranks <= 0 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return 4;   /* ncclInvalidArgument */
  id.internal[sizeof id.internal - 1] = 0;
  if (id.internal[0] != '/') return 4;
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 2;   /* ncclSystemError */
  if (ftruncate(fd, sizeof(fakeShared)) != 0) { close(fd); return 2; }
  fakeShared* sh = (fakeShared*)mmap(NULL, sizeof(fakeShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (sh == MAP_FAILED) return 2;
  fakeComm* c = (fakeComm*)calloc(1, sizeof *c);
  c->sh = sh; c->rank = rank; c->nranks = nranks;
  snprintf(c->name, sizeof c->name, "%s", id.internal);
  __sync_add_and_fetch(&sh->attached, 1);
  if (fake_barrier(c)) { munmap(sh, sizeof(fakeShared)); free(c); return 2; }   /* like RCCL: init returns once every rank is in */
  *comm = c;
  return 0;
}

/* in place or out of place, device pointers, fp64 (datatype 8) sum (op 0) only */
int ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
  fakeComm* c = (fakeComm*)comm;
  if (!c || datatype != 8 || op != 0 || count > FAKE_MAX_COUNT) return 4;
  if (c->sh->aborted) return 6;   /* ncclRemoteError */
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;   /* ncclUnhandledCudaError */
  if (hipMemcpy((void*)c->sh->slot[c->rank], sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  __sync_synchronize();
  if (fake_barrier(c)) return 6;
  double* sum = (double*)malloc(count * sizeof(double));
  for (size_t i = 0; i < count; ++i) {
    double a = 0.0;
    for (int r = 0; r < c->nranks; ++r) a += c->sh->slot[r][i];   /* rank order: the same bits on every rank */
    sum[i] = a;
  }
  if (fake_barrier(c)) { free(sum); return 6; }   /* nobody overwrites a slot before everybody has read it */
  if (c->rank == 0) __sync_add_and_fetch(&c->sh->calls, 1);
  hipError_t e = hipMemcpy(recvbuff, sum, count * sizeof(double), hipMemcpyHostToDevice);
  free(sum);
  return e == hipSuccess ? 0 : 1;
}

static int fake_release(fakeComm* c, int abort_flag) {
  if (!c) return 0;
  if (abort_flag) c->sh->aborted = 1;
  if (__sync_sub_and_fetch(&c->sh->attached, 1) <= 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(fakeShared));
  free(c);
  return 0;
}
int ncclCommDestroy(void* comm) { return fake_release((fakeComm*)comm, 0); }
int ncclCommAbort(void* comm) { return fake_release((fakeComm*)comm, 1); }
const char* ncclGetErrorString(int code) { (void)code; return "fake_rccl error"; }
/* test hook: completed all-reduces of this communicator's job */
int fakeRcclCalls(void* comm) { return comm ? ((fakeComm*)comm)->sh->calls : -1; }

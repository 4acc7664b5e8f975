// Probe: one-way latency of a self-flagging 8-byte word between two workgroups of one launch on gfx950, by ping-pong.
//   hipcc --offload-arch=gfx950 -O3 -o pingpong_probe pingpong_probe.hip && ./pingpong_probe
// Workgroup A stores round r into word 0 and polls word 1 for r; workgroup B polls word 0 and answers in word 1 (the two
// words lie in different 128-byte lines).  All other workgroups of the launch idle (they exit at once), so the pair can be
// chosen by its position: blockIdx % 8 is the XCD (round-robin dispatch, printed from HW_REG_XCC_ID).
// Flavours: 0 relaxed agent-scope atomic store / load (what hip/trdp.hip uses), 1 system-scope, 2 agent-scope store with
// an agent-scope atomic exchange as the poll (read-modify-write executes at the coherence point), 3 agent-scope atomic
// exchange as the STORE and agent-scope loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int F>
__device__ __forceinline__ void put(u64 *p, u64 v) {
  if (F == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if (F == 3) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int F>
__device__ __forceinline__ u64 get(u64 *p) {
  if (F == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (F == 2) return __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int F>
__global__ void k_pingpong(u64 *w, int a, int b, int rounds, u64 *out) {
  const int id = blockIdx.x;
  if (threadIdx.x != 0) return;
  if (id != a && id != b) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  out[id == a ? 2 : 3] = xcc & 15u;
  u64 *mine = w + (id == a ? 0 : 32), *other = w + (id == a ? 32 : 0);
  const u64 t0 = wall_clock64();
  long long spins = 0;
  for (int r = 1; r <= rounds; r++) {
    if (id == a) put<F>(mine, (u64)r);
    while (get<F>(other) != (u64)r) {
      if (++spins > (1ll << 26)) { out[4] = 1; return; }
    }
    if (id == b) put<F>(mine, (u64)r);
  }
  const u64 t1 = wall_clock64();
  if (id == a) { out[0] = t1 - t0; out[1] = (u64)spins; }
}

template <int F>
static void run(const char *name, u64 *w, u64 *out, int a, int b) {
  const int rounds = 2000;
  CK(hipMemset(w, 0, 4096));
  CK(hipMemset(out, 0, 64));
  hipLaunchKernelGGL(k_pingpong<F>, dim3(256), dim3(64), 0, 0, w, a, b, rounds, out);
  CK(hipDeviceSynchronize());
  u64 h[8];
  CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
  printf("%-28s workgroups %3d (XCC %llu) <-> %3d (XCC %llu): round trip %.3f us, one way %.3f us, %.1f polls per round%s\n", name, a, h[2], b, h[3],
         (double)h[0] * 0.01 / rounds, (double)h[0] * 0.005 / rounds, (double)h[1] / rounds, h[4] ? "  SPIN LIMIT" : "");
}

int main() {
  u64 *w, *out;
  CK(hipMalloc(&w, 4096));
  CK(hipMalloc(&out, 64));
  const int pairs[][2] = {{0, 8}, {0, 16}, {0, 1}, {0, 4}, {3, 7}, {8, 200}};
  for (auto &p : pairs) {
    run<0>("agent store / agent load", w, out, p[0], p[1]);
    run<1>("system store / system load", w, out, p[0], p[1]);
    run<2>("agent store / agent rmw poll", w, out, p[0], p[1]);
    run<3>("agent xchg / agent load", w, out, p[0], p[1]);
  }
  return 0;
}

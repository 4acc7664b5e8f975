// Probe: cost of one all-to-all exchange round between the workgroups of a persistent kernel on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o exchange_probe exchange_probe.hip && ./exchange_probe
// NG groups of NW workgroups (one group = one symmetry block of the eigensolve); per round every workgroup publishes
// its slice of a vector q (n values) and one workgroup (the "row owner", rotating) a whole vector z; every thread then
// needs q and z at its own three column indices.  Three protocols:
//   tagged : each double travels as two 8-byte words {32 data bits, 32-bit round tag}, written and read with relaxed
//            agent-scope atomics -- no fence, no flag, the data are their own flag (8-byte atomicity is all it needs);
//   barrier: plain stores, release fence, one counter per group, spin, acquire fence, plain loads;
//   tag-l2 : as tagged with WORKGROUP-scope accesses, for groups confined to one XCD -- does NOT work (measured: the
//            loads are served by the CU's L1 and never see another CU's stores; the run ends through its spin bound).
// Placement: "spread" = group g takes the workgroup ids g NW .. (ids are dealt round-robin over the 8 XCDs, so every group
// lives on all of them); "xcd" = group g takes the ids with id % 8 in {2g, 2g+1} (two XCDs per group); "one" = group g
// takes the ids with id % 8 == g (one XCD per group).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int NT = 512, CPT = 3;  // threads, columns per thread
constexpr int SPIN_MAX = 1 << 20;

// SCOPE: agent = coherent over the whole device (the multi-XCD L2s are not coherent with each other, so these accesses go
// to the memory side); workgroup = bypasses only the CU's L1 and is served by the XCD's own L2 -- outside the memory
// model between different workgroups, probed here for groups confined to ONE XCD (placement "one"), whose CUs share that
// L2.  The tags make a stale read harmless (it is simply polled again).
template <int SCOPE>
__device__ __forceinline__ void put_tagged(u64 *slot, double v, unsigned tag) {
  const u64 b = (u64)__double_as_longlong(v);
  __hip_atomic_store(slot, (b & 0xffffffffull) | ((u64)tag << 32), __ATOMIC_RELAXED, SCOPE);
  __hip_atomic_store(slot + 1, (b >> 32) | ((u64)tag << 32), __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ bool get_tagged(const u64 *slot, unsigned tag, double &v) {
  const u64 a = __hip_atomic_load(slot, __ATOMIC_RELAXED, SCOPE);
  const u64 b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, SCOPE);
  v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
  return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}

struct Args {
  u64 *tq, *tz;      // tagged: [group][parity][n][2]
  double *pq, *pz;   // barrier: [group][parity][n]
  unsigned *counter;  // [group]
  int *status;
  double *out;  // [group][NW]
  int n, NW, NG, rounds, placement;
};

__device__ __forceinline__ bool locate(const Args &a, int &g, int &w) {
  const int id = blockIdx.x;
  if (a.placement == 0) {
    g = id / a.NW;
    w = id % a.NW;
    return g < a.NG;
  }
  const int xcd = id & 7, slot = id >> 3;
  if (a.placement == 2) {  // one XCD per group
    g = xcd;
    w = slot;
    return g < a.NG && slot < a.NW;
  }
  const int per = a.NW / 2;
  g = xcd >> 1;
  w = (xcd & 1) * per + slot;
  return g < a.NG && slot < per;
}

template <int SCOPE>
__global__ __launch_bounds__(NT) void k_tagged(Args a) {
  int g, w;
  if (!locate(a, g, w)) return;
  const int n = a.n, sl = (n + a.NW - 1) / a.NW, t = threadIdx.x;
  double acc = 0.0;
  for (int r = 1; r <= a.rounds; r++) {
    u64 *q = a.tq + ((size_t)(g * 2 + (r & 1)) * n) * 2, *z = a.tz + ((size_t)(g * 2 + (r & 1)) * n) * 2;
    if (t < sl && w * sl + t < n) put_tagged<SCOPE>(q + 2 * (w * sl + t), 1.0 * r + 1e-3 * (w * sl + t) + acc * 1e-30, (unsigned)r);
    if (w == r % a.NW)
      for (int c = t; c < n; c += NT) put_tagged<SCOPE>(z + 2 * c, 2.0 * r + 1e-3 * c, (unsigned)r);
    for (int u = 0; u < CPT; u++) {
      const int c = t + NT * u;
      if (c >= n) continue;
      double vq, vz;
      int spins = 0;
      while (!get_tagged<SCOPE>(q + 2 * c, (unsigned)r, vq))
        if (++spins > SPIN_MAX || *(volatile int *)a.status) { atomicExch(a.status, 1); return; }
      while (!get_tagged<SCOPE>(z + 2 * c, (unsigned)r, vz))
        if (++spins > SPIN_MAX || *(volatile int *)a.status) { atomicExch(a.status, 1); return; }
      acc += vq + vz;
    }
    // a workgroup-wide dependency like the kernel's reductions: nobody publishes round r+1 before all its threads have round r
    __syncthreads();
  }
  // sum over the workgroup
  __shared__ double red[NT];
  red[t] = acc;
  __syncthreads();
  for (int s = NT / 2; s; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) a.out[g * a.NW + w] = red[0];
}

__global__ __launch_bounds__(NT) void k_barrier(Args a) {
  int g, w;
  if (!locate(a, g, w)) return;
  const int n = a.n, sl = (n + a.NW - 1) / a.NW, t = threadIdx.x;
  double acc = 0.0;
  for (int r = 1; r <= a.rounds; r++) {
    double *q = a.pq + (size_t)(g * 2 + (r & 1)) * n, *z = a.pz + (size_t)(g * 2 + (r & 1)) * n;
    if (t < sl && w * sl + t < n) q[w * sl + t] = 1.0 * r + 1e-3 * (w * sl + t) + acc * 1e-30;
    if (w == r % a.NW)
      for (int c = t; c < n; c += NT) z[c] = 2.0 * r + 1e-3 * c;
    __syncthreads();
    if (t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(a.counter + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(a.counter + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r * a.NW))
        if (++spins > SPIN_MAX) { atomicExch(a.status, 1); break; }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (*(volatile int *)a.status) return;
    for (int u = 0; u < CPT; u++) {
      const int c = t + NT * u;
      if (c < n) acc += __builtin_nontemporal_load(q + c) + __builtin_nontemporal_load(z + c);
    }
  }
  __shared__ double red[NT];
  red[t] = acc;
  __syncthreads();
  for (int s = NT / 2; s; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) a.out[g * a.NW + w] = red[0];
}

int main() {
  const int n = 1470, NGmax = 8, rounds = 2000;
  Args a{};
  a.n = n;
  a.rounds = rounds;
  CK(hipMalloc(&a.tq, sizeof(u64) * NGmax * 2 * n * 2));
  CK(hipMalloc(&a.tz, sizeof(u64) * NGmax * 2 * n * 2));
  CK(hipMalloc(&a.pq, sizeof(double) * NGmax * 2 * n));
  CK(hipMalloc(&a.pz, sizeof(double) * NGmax * 2 * n));
  CK(hipMalloc(&a.counter, sizeof(unsigned) * 8));
  CK(hipMalloc(&a.status, sizeof(int)));
  CK(hipMalloc(&a.out, sizeof(double) * 1024));
  // (out: NG * NW <= 8 * 84 entries)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // expected checksum per group: sum over rounds and columns of (1 r + 1e-3 c) + (2 r + 1e-3 c), per workgroup; all equal
  double expect = 0.0;
  for (int r = 1; r <= rounds; r++)
    for (int c = 0; c < n; c++) expect += 3.0 * r + 2e-3 * c;
  struct Cfg { int NW, NG, placement; };
  const Cfg cfgs[] = {{64, 1, 0}, {64, 3, 0}, {64, 3, 1}, {84, 3, 0}, {32, 3, 1}, {32, 3, 0}, {16, 3, 0}, {8, 3, 0}, {32, 3, 2}, {16, 3, 2}, {32, 8, 2}};
  for (const Cfg &c : cfgs)
    for (int proto = 0; proto < 3; proto++) {
      if (proto == 2 && c.placement != 2) continue;  // workgroup-scope accesses only for groups confined to one XCD
      a.NW = c.NW;
      a.NG = c.NG;
      a.placement = c.placement;
      const int grid = c.placement == 0 ? c.NW * c.NG : (c.placement == 2 ? 8 * c.NW : 8 * (c.NW / 2));
      CK(hipMemset(a.tq, 0, sizeof(u64) * NGmax * 2 * n * 2));
      CK(hipMemset(a.tz, 0, sizeof(u64) * NGmax * 2 * n * 2));
      CK(hipMemset(a.counter, 0, sizeof(unsigned) * 8));
      CK(hipMemset(a.status, 0, sizeof(int)));
      CK(hipMemset(a.out, 0, sizeof(double) * 1024));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      void *args[] = {&a};
      // cooperative launch: the runtime refuses a grid that is not co-resident
      const void *fn = proto == 0 ? (const void *)k_tagged<__HIP_MEMORY_SCOPE_AGENT>
                                  : (proto == 1 ? (const void *)k_barrier : (const void *)k_tagged<__HIP_MEMORY_SCOPE_WORKGROUP>);
      CK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(NT), args, 0, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      int st = 0;
      std::vector<double> out(1024);
      CK(hipMemcpy(&st, a.status, sizeof(int), hipMemcpyDeviceToHost));
      CK(hipMemcpy(out.data(), a.out, sizeof(double) * 1024, hipMemcpyDeviceToHost));
      double worst = 0.0;
      for (int i = 0; i < c.NW * c.NG; i++) worst = std::max(worst, std::abs(out[i] - expect) / expect);
      printf("%-7s NW=%3d NG=%d placement=%-6s grid=%3d: %7.3f us per round, status %d, checksum rel. error %.1e\n", proto == 0 ? "tagged" : (proto == 1 ? "barrier" : "tag-l2"),
             c.NW, c.NG, c.placement == 0 ? "spread" : (c.placement == 1 ? "xcd" : "one"), grid, 1e3 * ms / rounds, st, worst);
      fflush(stdout);
    }
  return 0;
}

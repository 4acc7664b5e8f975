// Lane maps of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction; one double per lane for
// each of a, b and the accumulator), found with one-hot data:  for every a-lane la, a = [lane == la], b = lane + 1;
// the non-zero result lanes ld then hold b[lb] of the b-lane that is multiplied with a[la] into d[ld].
// Printed: for each la the list of (ld <- lb).  profiles/r02_mfma4x4_lane_maps.txt holds the output.
// hipcc --offload-arch=gfx950 -O3 mfma4x4_probe.hip -o mfma4x4_probe && ./mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double *D) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++) {
    double a = (lane == la) ? 1.0 : 0.0;
    double b = lane + 1.0;
    double c = 0.0;
    c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    D[la * 64 + lane] = c;
  }
}
int main() {
  double *d, h[64 * 64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // hypothesis: a-lane = 16 blk + 4 k + i, b-lane = 16 blk + 4 k + j, d-lane = 16 blk + 4 i + j  (and the variants)
  int hyp[4] = {1, 1, 1, 1};
  for (int la = 0; la < 64; la++) {
    printf("a-lane %2d:", la);
    for (int ld = 0; ld < 64; ld++)
      if (h[la * 64 + ld] != 0.0) {
        int lb = (int)h[la * 64 + ld] - 1;
        printf(" d%-2d<-b%-2d", ld, lb);
        const int blk = la >> 4;
        // variant 0: a = (blk, k = (la>>2)&3, i = la&3), b = (blk, k, j = lb&3), d = (blk, i, j) at 16 blk + 4 i + j
        {
          int k = (la >> 2) & 3, i = la & 3;
          if (!((lb >> 4) == blk && ((lb >> 2) & 3) == k && ld == 16 * blk + 4 * i + (lb & 3))) hyp[0] = 0;
          // variant 1: d at 16 blk + 4 j + i
          if (!((lb >> 4) == blk && ((lb >> 2) & 3) == k && ld == 16 * blk + 4 * (lb & 3) + i)) hyp[1] = 0;
        }
        // variant 2: a = (blk, i = (la>>2)&3, k = la&3), b = (blk, j = (lb>>2)&3, k = lb&3), d = 16 blk + 4 i + j
        {
          int i = (la >> 2) & 3, k = la & 3;
          if (!((lb >> 4) == blk && (lb & 3) == k && ld == 16 * blk + 4 * i + ((lb >> 2) & 3))) hyp[2] = 0;
          if (!((lb >> 4) == blk && (lb & 3) == k && ld == 16 * blk + 4 * ((lb >> 2) & 3) + i)) hyp[3] = 0;
        }
      }
    printf("\n");
  }
  printf("hypotheses (a-lane, b-lane -> d-lane), blk = lane >> 4:\n");
  printf("  [0] a(i = l&3, k = (l>>2)&3)  b(j = l&3, k = (l>>2)&3)  d = 16 blk + 4 i + j : %d\n", hyp[0]);
  printf("  [1] a(i = l&3, k = (l>>2)&3)  b(j = l&3, k = (l>>2)&3)  d = 16 blk + 4 j + i : %d\n", hyp[1]);
  printf("  [2] a(k = l&3, i = (l>>2)&3)  b(k = l&3, j = (l>>2)&3)  d = 16 blk + 4 i + j : %d\n", hyp[2]);
  printf("  [3] a(k = l&3, i = (l>>2)&3)  b(k = l&3, j = (l>>2)&3)  d = 16 blk + 4 j + i : %d\n", hyp[3]);
  return 0;
}

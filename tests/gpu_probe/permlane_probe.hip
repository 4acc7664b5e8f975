// Lane maps of v_permlane32_swap_b32 / v_permlane16_swap_b32 and of the DPP controls used by csrc/hip/wave.h, printed
// from the hardware (gfx950): hipcc --offload-arch=gfx950 permlane_probe.hip -o permlane_probe && ./permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  const int l = threadIdx.x;
  unsigned X = l, Y = 100 + l;
  auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
  out[l] = r[0];
  out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(X, Y, false, false);
  out[128 + l] = s[0];
  out[192 + l] = s[1];
  out[256 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x128, 0xf, 0xf, true);   // row_ror:8
  out[320 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x141, 0xf, 0xf, true);   // row_half_mirror
  out[384 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x140, 0xf, 0xf, true);   // row_mirror
  out[448 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x111, 0xf, 0xf, true);   // row_shr:1
  out[512 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x142, 0xa, 0xf, true);   // row_bcast:15, rows 1 and 3
  out[576 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x143, 0xc, 0xf, true);   // row_bcast:31, rows 2 and 3
  out[640 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0xb1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
  out[704 + l] = __builtin_amdgcn_update_dpp(-1, (int)X, 0x4e, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
}
int main() {
  int *d, h[768];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char *names[12] = {"permlane32_swap[0] (X=lane, Y=100+lane)", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]",
                           "dpp row_ror:8", "dpp row_half_mirror", "dpp row_mirror", "dpp row_shr:1 (bound_ctrl: 0 fill)",
                           "dpp row_bcast:15 row_mask 0xa (old = -1)", "dpp row_bcast:31 row_mask 0xc (old = -1)",
                           "dpp quad_perm [1,0,3,2]", "dpp quad_perm [2,3,0,1]"};
  for (int q = 0; q < 12; q++) {
    printf("%s\n ", names[q]);
    for (int l = 0; l < 64; l++) printf(" %d", h[64 * q + l]);
    printf("\n");
  }
  return 0;
}

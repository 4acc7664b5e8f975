#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D, int* rowmap){
  // A: 16x4 (row-major A[i*4+k]), B: 4x16 (B[k*16+j])
  int lane = threadIdx.x;
  double a = A[(lane&15)*4 + (lane>>4)];
  double b = B[(lane>>4)*16 + (lane&15)];
  double4_t c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for(int r=0;r<4;r++){ D[(lane*4+r)] = c[r]; }
}
int main(){
  double hA[64], hB[64], hD[256];
  for(int i=0;i<16;i++) for(int k=0;k<4;k++) hA[i*4+k] = (i+1)*100 + k;   // asymmetric
  for(int k=0;k<4;k++) for(int j=0;j<16;j++) hB[k*16+j] = (k==0? (j+1): 0); // picks A[i][0]*(j+1)
  double *dA,*dB,*dD; int* dm;
  hipMalloc(&dA,sizeof(hA)); hipMalloc(&dB,sizeof(hB)); hipMalloc(&dD,sizeof(hD)); hipMalloc(&dm,4);
  hipMemcpy(dA,hA,sizeof(hA),hipMemcpyHostToDevice); hipMemcpy(dB,hB,sizeof(hB),hipMemcpyHostToDevice);
  probe<<<1,64>>>(dA,dB,dD,dm);
  hipMemcpy(hD,dD,sizeof(hD),hipMemcpyDeviceToHost);
  // expected C[i][j] = (i+1)*100*(j+1)
  int ok1=1, ok2=1;
  for(int lane=0;lane<64;lane++) for(int r=0;r<4;r++){
    double v=hD[lane*4+r];
    int j=lane&15; int i1=(lane>>4)+4*r; int i2=(lane>>4)*4+r;
    if(v != (i1+1)*100.0*(j+1)) ok1=0;
    if(v != (i2+1)*100.0*(j+1)) ok2=0;
  }
  printf("layout row=(lane>>4)+4*r : %d ; row=(lane>>4)*4+r : %d\n", ok1, ok2);
  printf("lane0 regs: %g %g %g %g ; lane16: %g %g %g %g\n", hD[0],hD[1],hD[2],hD[3],hD[64],hD[65],hD[66],hD[67]);
  return 0;
}

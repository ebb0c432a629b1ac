// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which input lane's 4 bf16 land where).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* o, int ld) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int lane = threadIdx.x & 63;
  int p = lane & 15, grp = lane >> 4;
  // lane p of group supplies address of row (p>>2), col chunk (p&3) of a 4x16 block at col grp*16
  int addr = (p >> 2) * ld + grp * 16 + (p & 3) * 4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr));
  for (int j = 0; j < 4; ++j) o[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  int ld = 128;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, ld);
  unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int lane = 0; lane < 64; ++lane) {
    printf("lane %2d:", lane);
    for (int j = 0; j < 4; ++j) {
      int v = h[lane * 4 + j]; printf(" (r%d,c%d)", v / ld, v % ld);
      int exp = j * ld + (lane >> 4) * 16 + (lane & 15);
      if (v != exp) ok = 0;
    }
    printf("\n");
  }
  printf("MODEL %s: lane l elem j = block[row j][col l&15]\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}

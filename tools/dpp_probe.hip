// Probe: which source lane does each DPP control / permlane swap deliver? (developer tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int BANK = 0xF>
__device__ unsigned dpp(unsigned old, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANK, false);
}
__global__ void probe(unsigned *out) {
  const unsigned lane = threadIdx.x;
  unsigned v = lane;
  int c = 0;
  out[64 * c++ + lane] = dpp<0xB1>(777u, v);            // quad_perm [1,0,3,2]
  out[64 * c++ + lane] = dpp<0x4E>(777u, v);            // quad_perm [2,3,0,1]
  out[64 * c++ + lane] = dpp<0x1B>(777u, v);            // quad_perm [3,2,1,0]
  out[64 * c++ + lane] = dpp<0x141>(777u, v);           // row_half_mirror
  out[64 * c++ + lane] = dpp<0x140>(777u, v);           // row_mirror
  out[64 * c++ + lane] = dpp<0x128>(777u, v);           // row_ror:8
  {
    unsigned t = dpp<0x124, 0x5>(777u, v);              // row_ror:4 on banks 0,2
    t = dpp<0x12C, 0xA>(t, v);                          // row_ror:12 on banks 1,3
    out[64 * c++ + lane] = t;
  }
  {
    unsigned t = dpp<0x12C, 0x5>(777u, v);              // row_ror:12 on banks 0,2
    t = dpp<0x124, 0xA>(t, v);                          // row_ror:4 on banks 1,3
    out[64 * c++ + lane] = t;
  }
  {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    out[64 * c++ + lane] = r[0];
    out[64 * c++ + lane] = r[1];
  }
  {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[64 * c++ + lane] = r[0];
    out[64 * c++ + lane] = r[1];
  }
  out[64 * c++ + lane] = dpp<0x142>(777u, v);           // row_bcast15
  out[64 * c++ + lane] = dpp<0x143>(777u, v);           // row_bcast31
}
int main() {
  unsigned *d, h[64 * 16];
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[] = {"qp[1,0,3,2]", "qp[2,3,0,1]", "qp[3,2,1,0]", "half_mirror", "row_mirror", "row_ror:8",
                         "ror4@b02+ror12@b13", "ror12@b02+ror4@b13", "pl32swap[0]", "pl32swap[1]", "pl16swap[0]",
                         "pl16swap[1]", "row_bcast15", "row_bcast31"};
  for (int c = 0; c < 14; ++c) {
    printf("%-20s:", names[c]);
    for (int l = 0; l < 64; ++l) printf(" %u", h[64 * c + l]);
    printf("\n");
  }
  return 0;
}

// LDS read instruction forms for a 16-byte window per lane at an 8-byte lane stride (conv_big_split.hip's B fragments of the
// channel-pair planar image): cycles per wave-instruction-equivalent (16 bytes per lane) with 4 and 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/lds_read_forms.hip -o scratch/ubench/lds_read_forms && ./lds_read_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int FORM>
__global__ void k(unsigned long long* out, unsigned* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lanes 0-15 / 16-31 / 32-47 / 48-63: four windows rows 32 (mod 64) dwords apart, like the kernel's channel pairs
  unsigned base = ((lane >> 4) * 1056 + (wave & 3) * 4352) * 4;
  if (FORM == 0) base += (lane & 15) * 16;             // aligned b128, conflict free
  else if (FORM == 4) base += (lane & 15) * 8 + 8 + 2; // 2-byte misaligned
  else base += (lane & 15) * 8 + 8;                    // 8-byte aligned windows, stride 8 bytes
  u32x4 acc = {0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (FORM == 0 || FORM == 1 || FORM == 4) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "n"(j * 512));
      } else if (FORM == 2) {
        u32x2 a, b;
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=&v"(a), "=&v"(b) : "v"(base), "n"(j * 512), "n"(j * 512 + 8));
        v[j] = u32x4{a[0], a[1], b[0], b[1]};
      } else {
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v[j]) : "v"(base), "n"(j * 28), "n"(j * 28 + 1));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345) sink[0] = 1;
}

int main() {
  unsigned long long* out;
  unsigned* sink;
  hipMalloc(&out, 256 * 16 * 8);
  hipMalloc(&sink, 4);
  const int iters = 2000;
  const char* names[5] = {"ds_read_b128 aligned, stride 16", "ds_read_b128 8-byte aligned, stride 8", "2 x ds_read_b64, stride 8",
                          "ds_read2_b64 adjacent, stride 8", "ds_read_b128 2-byte misaligned, stride 8"};
  for (int waves = 4; waves <= 16; waves *= 2)
    for (int f = 0; f < 5; ++f) {
      void (*kern)(unsigned long long*, unsigned*, int) = f == 0 ? k<0> : f == 1 ? k<1> : f == 2 ? k<2> : f == 3 ? k<3> : k<4>;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 96 * 1024, 0, out, sink, iters);
      hipDeviceSynchronize();
      unsigned long long h[16];
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      double mx = 0;
      for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
      printf("%2d waves/CU  %-42s %6.1f clk per 16-byte wave read (per wave), %5.2f LDS clk per wave-instruction across the CU\n", waves,
             names[f], mx / (iters * 8.0), mx / (iters * 8.0) / waves);
    }
  return 0;
}

// Microbenchmark: are ds_read_b128 / ds_read_b64 at 2-byte-aligned LDS addresses correct on gfx950, and what do they cost?
// (Design question for a bf16-split weight-gradient: an A operand = 8 consecutive bf16 of an x row starting at an odd element.)
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/lds_unaligned tools/micro/lds_unaligned.hip ; ./tools/micro/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(2))) P16 { u32x4 v; };
struct __attribute__((packed, aligned(2))) P8 { u32x2 v; };

__global__ void check(u32x4* out, int s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (unsigned char)(i * 7);
  __syncthreads();
  out[threadIdx.x] = reinterpret_cast<const P16*>(sm + 16 * threadIdx.x + 2 * s)->v;
}

template <int BYTES>
__global__ __launch_bounds__(256) void timed(unsigned* out, int shift, int stride, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  for (int i = threadIdx.x; i < 65536; i += 256) sm[i] = (unsigned char)(i * 7);
  __syncthreads();
  unsigned acc = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int off = wave * 8192 + lane * stride + shift;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = (off + u * 1024) & 0xffff;
      if constexpr (BYTES == 16) { const u32x4 v = reinterpret_cast<const P16*>(sm + (o & 0xfffe))->v; acc += v.x ^ v.y ^ v.z ^ v.w; }
      else { const u32x2 v = reinterpret_cast<const P8*>(sm + (o & 0xfffe))->v; acc += v.x ^ v.y; }
    }
    off += 32;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  u32x4* d; hipMalloc(&d, 256 * 16);
  for (int s = 0; s < 8; ++s) {
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 8192 + 64, 0, d, s);
    u32x4 h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) for (int b = 0; b < 16; ++b) bad += (unsigned char)((16 * t + 2 * s + b) * 7) != ((unsigned char*)&h[t])[b];
    printf("ds_read_b128 at +%d bytes: %d wrong bytes\n", 2 * s, bad);
  }
  unsigned* o; hipMalloc(&o, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)timed<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)timed<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 2000;
  for (int bytes : {16, 8})
    for (int stride : {16, 8})
      for (int shift : {0, 2, 4, 6, 8}) {
        if (bytes == 16 && stride == 8) continue;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (bytes == 16) hipLaunchKernelGGL(timed<16>, dim3(256), dim3(256), 65536, 0, o, shift, stride, iters);
          else hipLaunchKernelGGL(timed<8>, dim3(256), dim3(256), 65536, 0, o, shift, stride, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep) printf("ds_read_b%d lane stride %2d B shift %d B: %.3f ms  -> %.1f cycles per wave-instruction per CU (at 2.4 GHz, 4 waves per CU)\n",
                          bytes * 8, stride, shift, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0 * 4));
        }
      }
  return 0;
}

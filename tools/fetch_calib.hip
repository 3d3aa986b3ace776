// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the two load paths the GEMM uses
// (MI355X_MICROARCH.md "HBM": calibrate on a known byte count in your own access pattern).
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib
// Both kernels stream the same 4 GiB once (one 1 KiB row per wave instruction).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void stream_reg(const double *src, size_t n2, double *sink) {
  v2d acc = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const v2d v = *reinterpret_cast<const v2d *>(src + 2 * i);
    acc += v;
  }
  if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void stream_glds(const double *src, size_t n2, double *sink) {
  __shared__ __attribute__((aligned(16))) double buf[4 * 128];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int wave = threadIdx.x >> 6;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    __builtin_amdgcn_global_load_lds((glb_void *)(src + 2 * i), (lds_void *)(buf + wave * 128), 16, 0, 0);
  __syncthreads();
  if (buf[threadIdx.x] == 12345.678) sink[0] = buf[0];
}

int main() {
  const size_t bytes = 4ull << 30, n2 = bytes / 16;
  double *src, *sink;
  hipMalloc(&src, bytes);
  hipMalloc(&sink, 64);
  hipMemset(src, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_reg, dim3(4096), dim3(256), 0, 0, src, n2, sink);
    hipLaunchKernelGGL(stream_glds, dim3(4096), dim3(256), 0, 0, src, n2, sink);
  }
  hipDeviceSynchronize();
  printf("streamed %zu bytes per kernel\n", bytes);
  return 0;
}

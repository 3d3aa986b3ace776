// Developer experiment: what brings a stream back after another thread's legacy-stream call invalidated its ThreadLocal capture?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <atomic>
__global__ void k(double *p) { p[threadIdx.x] += 1.0; }
static const char *st(hipStream_t s) {
  hipStreamCaptureStatus cs; hipError_t e = hipStreamIsCapturing(s, &cs); (void)hipGetLastError();
  static char b[64]; snprintf(b, sizeof b, "%s/status %d", hipGetErrorName(e), (int)cs); return b;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *d, *d2; hipMalloc(&d, 1024); hipMalloc(&d2, 1024); hipMemset(d, 0, 1024);
  double h[8];
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int mode = 0; mode < 2; ++mode) {
  std::atomic<int> phase{0};
  std::thread other([&] { while (phase.load() != 1) {} hipError_t e = hipMemcpy(h, d2, 64, hipMemcpyDeviceToHost); printf("other thread hipMemcpy during capture: %s\n", hipGetErrorName(e)); (void)hipGetLastError(); phase = 2; });
  hipError_t e = hipStreamBeginCapture(s, mode == 0 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed);
  printf("mode %s begin: %s  [%s]\n", mode == 0 ? "ThreadLocal" : "Relaxed", hipGetErrorName(e), st(s));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  phase = 1; while (phase.load() != 2) {}
  other.join();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  printf("launch after the intrusion: %s  [%s]\n", hipGetErrorName(hipGetLastError()), st(s));
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(s, &g);
  printf("EndCapture: %s graph %p [%s]\n", hipGetErrorName(e), (void *)g, st(s)); (void)hipGetLastError();
  e = hipStreamEndCapture(s, &g);
  printf("EndCapture again: %s [%s]\n", hipGetErrorName(e), st(s)); (void)hipGetLastError();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  printf("eager launch: %s [%s]\n", hipGetErrorName(hipGetLastError()), st(s));
  e = hipStreamSynchronize(s); printf("sync: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
  e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); printf("begin again: %s [%s]\n", hipGetErrorName(e), st(s)); (void)hipGetLastError();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  e = hipStreamEndCapture(s, &g); printf("end again: %s graph %p [%s]\n", hipGetErrorName(e), (void *)g, st(s)); (void)hipGetLastError();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
  printf("eager launch 2: %s [%s]\n", hipGetErrorName(hipGetLastError()), st(s));
  void *m = nullptr; e = hipMalloc(&m, 256); printf("hipMalloc on this thread: %s\n", hipGetErrorName(e)); (void)hipGetLastError();
  hipStream_t s2; e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); printf("new stream: %s\n", hipGetErrorName(e));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d); printf("launch on the new stream: %s\n", hipGetErrorName(hipGetLastError()));
  hipStreamSynchronize(s2);
  e = hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s2); printf("value %g (%s)\n\n", h[0], hipGetErrorName(e));
  }
  return 0;
}

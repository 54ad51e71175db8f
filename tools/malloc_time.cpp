// how long hipMalloc / hipMemset / hipFree of node-sized blocks take (tuning aid)
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
int main() {
  hipFree(0);
  for (size_t gb : {1, 4, 16, 30}) {
    void* p = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, gb << 30);
    auto t1 = std::chrono::steady_clock::now();
    hipMemset(p, 0, gb << 30); hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    hipMemset(p, 0xff, gb << 30); hipDeviceSynchronize();
    auto t3 = std::chrono::steady_clock::now();
    hipFree(p);
    auto t4 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    printf("%zu GB: malloc %.1f ms (rc %d), first memset %.1f ms, second memset %.1f ms, free %.1f ms\n", gb, ms(t0, t1), (int)e, ms(t1, t2), ms(t2, t3), ms(t3, t4));
  }
  return 0;
}

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <dlfcn.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  double t0 = now();
  hipFree(0);
  double t1 = now();
  void* h = dlopen(argv[1], RTLD_NOW);
  double t2 = now();
  if (!h) { printf("dlopen failed %s\n", dlerror()); return 1; }
  typedef int (*create_t)(int, void**);
  create_t cr = (create_t)dlsym(h, "dafs_hip_create");
  void* ctx = nullptr;
  int rc = cr ? cr(0, &ctx) : -99;
  double t3 = now();
  printf("hip init %.3f  dlopen %.3f  create %.3f rc=%d\n", t1 - t0, t2 - t1, t3 - t2, rc);
  return 0;
}

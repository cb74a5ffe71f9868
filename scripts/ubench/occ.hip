// occupancy of a 64-thread workgroup with a given dynamic LDS size (bytes): how many fit one CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(64, 2) k(float* out) { extern __shared__ float lds[]; lds[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = lds[63 - threadIdx.x]; }
int main(int argc, char** argv) {
  for (int i = 1; i < argc; i++) {
    int bytes = atoi(argv[i]), n = 0;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, 64, bytes);
    printf("lds %d bytes -> %d workgroups per CU (%s)\n", bytes, n, hipGetErrorString(e));
  }
  return 0;
}

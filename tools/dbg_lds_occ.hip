// probe: how many 64-thread workgroups fit on a CU as a function of the dynamic LDS size (allocation granularity)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1.f; o[threadIdx.x] = s[threadIdx.x ^ 1]; }
int main() {
    for (int bytes : {16384, 17408, 17728, 17920, 18176, 18204, 18432, 18944, 20480}) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, bytes);
        printf("dynamic LDS %6d B -> %d workgroups per CU\n", bytes, nb);
    }
    return 0;
}

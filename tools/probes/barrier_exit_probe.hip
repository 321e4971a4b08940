// Does s_barrier count only the SURVIVING waves of a workgroup?  (GCN/CDNA ISA: "if some waves in the threadgroup have
// already terminated, this waits on only the surviving waves".)  4-wave workgroups; `keep` waves stay and run `rounds`
// barrier-separated LDS exchanges, the others end right after the first barrier.  Also reports whether the wave slots of
// the ended waves are re-used by NEW workgroups while the survivors still run (resident workgroups per CU over time).
// Diagnostic only; run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int keep_light, int rounds, int spin) {
    __shared__ int box[4];
    __shared__ int sum;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) sum = 0;
    __syncthreads();
    const int keep = (blockIdx.x & 7) == 0 ? 4 : keep_light;     // one workgroup in eight keeps all its waves
    if (wave >= keep) return;
    int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        if (lane == 0) box[wave] = r * 10 + wave;
        __syncthreads();
        for (int k = 0; k < keep; ++k) acc += box[k];
        volatile float x = 1.0f;
        for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
        __syncthreads();
    }
    if (lane == 0) atomicAdd(&sum, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = (unsigned long long)sum;
        out[blockIdx.x * 4 + 1] = t0;
        out[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 3] = keep;
    }
}
int main(int argc, char** argv) {
    const int nb = 4096, rounds = 20;
    const int keep_light = argc > 1 ? atoi(argv[1]) : 2;
    const int spin = argc > 2 ? atoi(argv[2]) : 300;
    const int lds = argc > 3 ? atoi(argv[3]) : 8 * 1024;
    unsigned long long* d;
    hipMalloc(&d, nb * 4 * sizeof(unsigned long long));
    hipMemset(d, 0, nb * 4 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), lds, 0, d, keep_light, rounds, spin);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<unsigned long long> h(nb * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nb; ++b) {
        const int keep = (int)h[b * 4 + 3];
        long long want = 0;
        for (int r = 0; r < rounds; ++r)
            for (int k = 0; k < keep; ++k) want += r * 10 + k;
        want *= keep;
        if ((long long)h[b * 4] != want) ++bad;
        tmin = std::min(tmin, h[b * 4 + 1]);
        tmax = std::max(tmax, h[b * 4 + 2]);
    }
    printf("keep_light=%d lds=%d: wrong sums %d of %d; launch span %.1f us\n", keep_light, lds, bad, nb, (tmax - tmin) / 100.0);
    // resident workgroups over time
    const double span = (tmax - tmin) / 100.0;
    for (int k = 0; k < 10; ++k) {
        const unsigned long long t = tmin + (unsigned long long)((k + 0.5) * span * 10.0);
        int res = 0;
        for (int b = 0; b < nb; ++b) res += (h[b * 4 + 1] <= t && t < h[b * 4 + 2]) ? 1 : 0;
        printf("  t=%.1f us resident %d\n", (t - tmin) / 100.0, res);
    }
    return bad ? 1 : 0;
}

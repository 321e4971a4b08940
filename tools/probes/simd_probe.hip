// Where does the dispatcher put the waves of a 256-thread workgroup?  Prints, for each physical wave
// index 0..3, the histogram of SIMD ids (HW_REG_HW_ID bits 5:4), and how many distinct (xcc, se, cu)
// a run of consecutive block ids covers.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    volatile float x = 1.0f;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;   // keep blocks resident for a while
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (x == 123.f) smem[threadIdx.x] = 1;
}
int main() {
    const int nb = 4096;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
    probe<<<nb, 256, 11 * 1024>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    std::map<unsigned, std::vector<int>> per_cu;
    for (int b = 0; b < nb; ++b) {
        for (int w = 0; w < 4; ++w) hist[w][(h[(b * 4 + w) * 2] >> 4) & 3]++;
        unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 0xF;
        unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 12);
        per_cu[key].push_back(b);
    }
    for (int w = 0; w < 4; ++w) printf("physical wave %d -> simd histogram: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("distinct CUs seen: %zu\n", per_cu.size());
    int shown = 0;
    for (auto& kv : per_cu) {
        if (shown++ >= 4) break;
        printf("cu key %05x blocks:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    // first 16 blocks: xcc ids
    printf("xcc of blocks 0..15:");
    for (int b = 0; b < 16; ++b) printf(" %u", h[b * 8 + 1] & 0xF);
    printf("\n");
    return 0;
}

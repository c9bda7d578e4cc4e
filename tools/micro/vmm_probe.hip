// probe of the virtual-memory calls the growable chunk pool uses: hipcc --offload-arch=gfx950 -o vmm_probe vmm_probe.hip && ./vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
static bool g_ok = true;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_ok = false; (void)hipGetLastError(); } printf("%-78s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
__global__ void touch(float *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0f; }
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity min %zu recommended %zu\n", gmin, grec);
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t gran = 2u << 20, reserve = 16 * gran;
    for (int mode = 0; mode < 3; mode++) {
        printf("== mode %d (0: access on the new range, 1: access on the whole mapped range, 2: as 0 but the device idle)\n", mode);
        g_ok = true;
        char *base[3] = {nullptr, nullptr, nullptr};
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (int a = 0; a < 3; a++) {
            CK(hipMemAddressReserve((void **)&base[a], reserve, gran, nullptr, 0));
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, gran, &prop, 0)); hs.push_back(h);
            CK(hipMemMap(base[a], gran, 0, h, 0));
            CK(hipMemSetAccess(base[a], gran, &acc, 1));
        }
        size_t mapped = gran;
        for (int step = 0; step < 3; step++) {
            if (g_ok) for (int a = 0; a < 3; a++) touch<<<(unsigned)((mapped / 4 + 255) / 256), 256, 0, st>>>((float *)base[a], mapped / 4);
            if (mode == 2) CK(hipStreamSynchronize(st));
            const size_t add = mapped;  // double
            for (int a = 0; a < 3; a++) {
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, add, &prop, 0)); hs.push_back(h);
                CK(hipMemMap(base[a] + mapped, add, 0, h, 0));
                if (mode == 1) CK(hipMemSetAccess(base[a], mapped + add, &acc, 1));
                else CK(hipMemSetAccess(base[a] + mapped, add, &acc, 1));
            }
            mapped += add;
        }
        if (g_ok) for (int a = 0; a < 3; a++) touch<<<(unsigned)((mapped / 4 + 255) / 256), 256, 0, st>>>((float *)base[a], mapped / 4);
        CK(hipStreamSynchronize(st));
        for (int a = 0; a < 3; a++) { CK(hipMemUnmap(base[a], mapped)); }
        for (auto h : hs) (void)hipMemRelease(h);
        for (int a = 0; a < 3; a++) CK(hipMemAddressFree(base[a], reserve));
    }
    return 0;
}

// What does hipMemUnmap do with a range that spans several mappings?  (round 3: a workspace of mapped chunks was
// released by ONE hipMemUnmap over all of them; a later run faulted.)  hipcc -o /tmp/vmm_probe vmm_unmap_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
int main() {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    printf("granularity rc=%d", (int)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf(" gran=%zu\n", gran);
    const size_t chunk = (size_t)1 << 28, nch = 4, total = chunk * nch;
    for (int mode = 0; mode < 2; mode++) {
        void *va = nullptr;
        printf("mode %d: reserve rc=%d\n", mode, (int)hipMemAddressReserve(&va, total, chunk, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t i = 0; i < nch; i++) {
            hipMemGenericAllocationHandle_t h;
            int r1 = (int)hipMemCreate(&h, chunk, &prop, 0);
            int r2 = (int)hipMemMap((char *)va + i * chunk, chunk, 0, h, 0);
            if (r1 || r2) printf("  create/map %zu: %d %d\n", i, r1, r2);
            hs.push_back(h);
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        printf("  set access rc=%d\n", (int)hipMemSetAccess(va, total, &acc, 1));
        printf("  memset rc=%d sync rc=%d\n", (int)hipMemset(va, 1, total), (int)hipDeviceSynchronize());
        if (mode == 0) {
            printf("  ONE unmap over all %zu mappings: rc=%d\n", nch, (int)hipMemUnmap(va, total));
            (void)hipGetLastError();
            for (size_t i = 0; i < nch; i++) printf("  then unmap chunk %zu alone: rc=%d\n", i, (int)hipMemUnmap((char *)va + i * chunk, chunk));
        } else {
            for (size_t i = 0; i < nch; i++) printf("  unmap chunk %zu: rc=%d\n", i, (int)hipMemUnmap((char *)va + i * chunk, chunk));
        }
        for (size_t i = 0; i < nch; i++) { int r = (int)hipMemRelease(hs[i]); if (r) printf("  release %zu rc=%d\n", i, r); }
        printf("  address free rc=%d\n", (int)hipMemAddressFree(va, total));
    }
    return 0;
}

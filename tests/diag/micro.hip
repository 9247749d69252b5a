// Diagnostic micro-benchmarks of cm_core.h bodies (not part of the product, not a test).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icircminer_amd/csrc tests/diag/micro.hip -o /tmp/micro && /tmp/micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "circminer_hot.h"
#include "cm_core.h"
using namespace cmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_bsearch(KCore kc, const uint32_t *pos, int reps, int *out) {
    const Core c = to_core(kc);
    const int t = blockIdx.x * 64 + threadIdx.x;
    int acc = 0;
    uint32_t p = pos[t];
    for (int r = 0; r < reps; ++r) { int ind; acc += overlap_ind(c, p + (uint32_t)(r & 3), ind) + ind; }
    out[t] = acc;
}
__global__ void k_dp(KCore kc, const uint32_t *gpos, const uint8_t *reads, int rl, int n, int m, int reps, int mode, int str_cap, int *out) {
    extern __shared__ uint32_t lds_words[];
    CM_L uint8_t *lane_base = (CM_L uint8_t *)lds_words + 4 * threadIdx.x;
    const Core c = to_core(kc);
    const int t = blockIdx.x * 64 + threadIdx.x;
    int err = 0;
    const DpMem sm{LBuf{lane_base, str_cap}, LBuf{lane_base + str_cap * 64, str_cap}, (g_err)out};
    const SV ref{c.X.genome, (int32_t)gpos[t], 1, 0};
    const SV q{(g_u8)(reads + (size_t)t * rl), 0, 1, 0};
    int acc = 0;
    for (int r = 0; r < reps; ++r) {
        int a, b, s;
        if (mode == 0) acc += local_alignment_sc(c, sm, ref, n, q, m, a, b, s) + s;
        else if (mode == 1) acc += local_alignment_side(c, sm, q, n, ref, m, false, a, s) + s;
        else { stage(ref, n, sm.a, 4); stage(q, m, sm.b, 5); acc += sm.a.get(r % n) + sm.b.get(r % m); }
    }
    out[t + 64] = acc + err;
}

int main() {
    const int NW = 4096, NT = NW * 64;
    std::mt19937 rng(1);
    const uint32_t L = 4000000;
    std::vector<uint8_t> g(L + 128);
    for (auto &x : g) x = "ACGT"[rng() & 3];
    const int NIV = 2048;
    std::vector<uint32_t> sp(NIV), ep(NIV), zero(NIV + 1, 0), segoff(NIV + 1), seg(NIV);
    for (int i = 0; i < NIV; ++i) { sp[i] = 1000 + i * 1900; ep[i] = sp[i] + 300; segoff[i] = i; seg[i] = i; }
    segoff[NIV] = NIV;
    KCore kc{};
    kc.P = cm_params{20, 500, 300, 0, 4, 7, 3, 500, 2000000, 30, 0, 0};
    uint8_t *dg; CK(hipMalloc(&dg, g.size())); CK(hipMemcpy(dg, g.data(), g.size(), hipMemcpyHostToDevice));
    kc.X.genome = dg + 64; kc.X.ref_len = L;
    uint32_t *dsp, *dep, *dso, *dsg;
    CK(hipMalloc(&dsp, NIV * 4)); CK(hipMalloc(&dep, NIV * 4)); CK(hipMalloc(&dso, (NIV + 1) * 4)); CK(hipMalloc(&dsg, NIV * 4));
    CK(hipMemcpy(dsp, sp.data(), NIV * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dep, ep.data(), NIV * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dso, segoff.data(), (NIV + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsg, seg.data(), NIV * 4, hipMemcpyHostToDevice));
    kc.A.n_iv = NIV; kc.A.iv_spos = dsp; kc.A.iv_epos = dep; kc.A.iv_seg_off = dso; kc.A.iv_seg = dsg;
    int *dout; CK(hipMalloc(&dout, (NT + 64) * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, int reps) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // 4096 waves at 1024 resident (if occupancy 1) -> report wave-time = ms * resident / NW
        printf("%-44s %8.3f ms  -> %.2f us per call per wave (assuming 1 wave/SIMD: x1024/%d)\n", name, ms, ms * 1000.0 * 1024 / NW / reps, NW);
    };
    for (int distinct = 0; distinct < 2; ++distinct) {
        std::vector<uint32_t> pos(NT), gp(NT);
        const int RL = 160;
        std::vector<uint8_t> rd((size_t)NT * RL);
        for (int t = 0; t < NT; ++t) {
            const int src = distinct ? t : (t / 64) * 64;           // identical within a wave vs all different
            std::mt19937 r2(src * 7919 + 13);
            pos[t] = 1000 + (r2() % (NIV * 1900));
            gp[t] = 5000 + (r2() % (L - 10000));
            for (int i = 0; i < RL; ++i) rd[(size_t)t * RL + i] = g[64 + gp[t] + i];
            if (r2() & 1) rd[(size_t)t * RL + 5 + (r2() % 100)] = 'A';   // a mismatch or not
        }
        uint32_t *dpos, *dgp; uint8_t *drd;
        CK(hipMalloc(&dpos, NT * 4)); CK(hipMalloc(&dgp, NT * 4)); CK(hipMalloc(&drd, rd.size() + 128));
        CK(hipMemcpy(dpos, pos.data(), NT * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgp, gp.data(), NT * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(drd + 64, rd.data(), rd.size(), hipMemcpyHostToDevice));
        printf("---- lanes of a wave: %s\n", distinct ? "ALL DIFFERENT" : "IDENTICAL");
        timeit("binary search x16", [&] { hipLaunchKernelGGL(k_bsearch, dim3(NW), dim3(64), 0, 0, kc, dpos, 16, dout); }, 16);
        const int cap = 320;
        for (int mode = 0; mode < 3; ++mode)
            for (int len : {10, 130}) {
                char nm[96];
                snprintf(nm, sizeof nm, "%s n=%d m=%d x4", mode == 0 ? "stage+xdrop" : mode == 1 ? "stage+edit" : "stage only", len + 3, len);
                const int n = mode == 1 ? len + 3 : len + 3, m = len;
                timeit(nm, [&] { hipLaunchKernelGGL(k_dp, dim3(NW), dim3(64), 2 * cap * 64, 0, kc, dgp, drd + 64, RL, n, m, 4, mode, cap, dout); }, 4);
            }
        CK(hipFree(dpos)); CK(hipFree(dgp)); CK(hipFree(drd));
    }
    return 0;
}

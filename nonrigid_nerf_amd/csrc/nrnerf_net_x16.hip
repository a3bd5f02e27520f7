// nrnerf_net_x16.hip -- instantiations of the 16x16x32 trunk-only kernel (nrnerf_net_x16.h): compiled architecture 0's trunk, bf16 and
// f16, one object per compositing case (-DNRN_X16_EPL=0..4: raw outputs to memory / compositing fused in for passes of up to 64, 128,
// 192, 256 samples per ray); the object of case 0 also holds the dispatcher.
#include "nrnerf_net_x16.h"
#include "nrnerf_x16_api.h"

#ifndef NRN_X16_EPL
#error "compile with -DNRN_X16_EPL=0..4 (Makefile)"
#endif
#define NRN_CAT2(a, b) a##b
#define NRN_CAT(a, b) NRN_CAT2(a, b)
// -DNRN_X16_SAMPLE=1 (with EPL 1 or 2: coarse passes of up to 64 / 128 samples): the objects with the coarse epilogue -- compositing +
// sample_pdf + merge behind the trunk (launch_net_x16_s1 / _s2)
#ifndef NRN_X16_SAMPLE
#define NRN_X16_SAMPLE 0
#endif
#if NRN_X16_SAMPLE
#define NRN_X16_ENTRY NRN_CAT(launch_net_x16_s, NRN_X16_EPL)
#else
#define NRN_X16_ENTRY NRN_CAT(launch_net_x16_e, NRN_X16_EPL)
#endif

namespace nrn {
// arch: 0 = the default trunk (8 x 256), 5 = --netwidth 128 (ArchNarrow; the ids of nrnerf_net.hip's dispatch table)
hipError_t NRN_X16_ENTRY(int precision, int arch, bool views, const NetArgs& a, int num_cus, hipStream_t stream) {
    constexpr bool SMP = NRN_X16_SAMPLE != 0;
    if (views) {
        if (arch != 0) return hipErrorInvalidValue;
        if (precision == PREC_BF16) return launch_net_x16_t<PolBF16, ArchDefault, NRN_X16_EPL, true, SMP>(a, num_cus, stream);
        if (precision == PREC_F16) return launch_net_x16_t<PolF16, ArchDefault, NRN_X16_EPL, true, SMP>(a, num_cus, stream);
        return hipErrorInvalidValue;
    }
    if (arch == 0) {
        if (precision == PREC_BF16) return launch_net_x16_t<PolBF16, ArchDefault, NRN_X16_EPL, false, SMP>(a, num_cus, stream);
        if (precision == PREC_F16) return launch_net_x16_t<PolF16, ArchDefault, NRN_X16_EPL, false, SMP>(a, num_cus, stream);
    } else if (arch == 5) {
        if (precision == PREC_BF16) return launch_net_x16_t<PolBF16, ArchNarrow, NRN_X16_EPL, false, SMP>(a, num_cus, stream);
        if (precision == PREC_F16) return launch_net_x16_t<PolF16, ArchNarrow, NRN_X16_EPL, false, SMP>(a, num_cus, stream);
    }
    return hipErrorInvalidValue;
}

#if NRN_X16_EPL == 0 && !NRN_X16_SAMPLE
// rays of one fused-compositing group (one workgroup iteration set): WAVES waves x the fewest rays whose 16-sample blocks fill whole
// iterations of NB blocks -- the API layer's "enough rays to fuse" threshold asks here instead of restating the kernel's mapping
long long x16_rays_per_group(int arch, int S) {
    const int bpr = (S + 15) / 16;
    const int NB = (arch == 5) ? X16Cfg<ArchNarrow>::NB : X16Cfg<ArchDefault>::NB, WAVES = (arch == 5) ? X16Cfg<ArchNarrow>::WAVES : X16Cfg<ArchDefault>::WAVES;
    const int RW = (bpr % NB == 0) ? 1 : ((2 * bpr) % NB == 0 ? 2 : NB);
    return (long long)WAVES * RW;
}
hipError_t launch_net_x16_e1(int, int, bool, const NetArgs&, int, hipStream_t);
hipError_t launch_net_x16_e2(int, int, bool, const NetArgs&, int, hipStream_t);
hipError_t launch_net_x16_e3(int, int, bool, const NetArgs&, int, hipStream_t);
hipError_t launch_net_x16_e4(int, int, bool, const NetArgs&, int, hipStream_t);
hipError_t launch_net_x16_s1(int, int, bool, const NetArgs&, int, hipStream_t);
hipError_t launch_net_x16_s2(int, int, bool, const NetArgs&, int, hipStream_t);
int x16_coarse_epilogue_max_samples() { return 128; }          // (which coarse passes have an object with the epilogue: EPL 1 and 2)
hipError_t launch_net_x16(int precision, int arch, bool views, const NetArgs& a, int num_cus, hipStream_t stream) {
    if (!a.fuse_on) return launch_net_x16_e0(precision, arch, views, a, num_cus, stream);
    if (a.fuse.n_importance > 0) {          // the coarse pass of a hierarchical render with its epilogue
        switch ((a.S + 63) / 64) {
            case 1: return launch_net_x16_s1(precision, arch, views, a, num_cus, stream);
            case 2: return launch_net_x16_s2(precision, arch, views, a, num_cus, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch ((a.S + 63) / 64) {
        case 1: return launch_net_x16_e1(precision, arch, views, a, num_cus, stream);
        case 2: return launch_net_x16_e2(precision, arch, views, a, num_cus, stream);
        case 3: return launch_net_x16_e3(precision, arch, views, a, num_cus, stream);
        case 4: return launch_net_x16_e4(precision, arch, views, a, num_cus, stream);
        default: return hipErrorInvalidValue;
    }
}
#endif
}  // namespace nrn

#if defined(NRN_TIMING) && NRN_X16_EPL == 3 && !NRN_X16_SAMPLE
// reads and clears the per-phase cycle counters of the 129..192-sample kernel (tools/timing_probe.py --x16): out[8 waves][8 slots]
extern "C" int nrnerf_debug_timing_launch_net_x16(unsigned long long* out) {
    static const unsigned long long zero[64] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nrn::g_nrn_timing), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(nrn::g_nrn_timing), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

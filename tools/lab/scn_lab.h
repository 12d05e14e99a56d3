// tools/lab/scn_lab.h -- the LAB's copy of scnerf_amd/csrc/device/scn_lab.h: the switches of the timing experiments set
// from -DSCN_H3_<FLAG> (tools/ablate_h3.sh puts this directory first on the include path).  Never part of the product build.
#pragma once
namespace scn {
namespace lab {
#ifdef SCN_H3_NO_EPI
constexpr bool kNoEpilogue = true;
#else
constexpr bool kNoEpilogue = false;
#endif
#ifdef SCN_H3_NO_STORE
constexpr bool kNoStore = true;
#else
constexpr bool kNoStore = false;
#endif
#ifdef SCN_H3_NO_PESTORE
constexpr bool kNoPeStore = true;
#else
constexpr bool kNoPeStore = false;
#endif
#ifdef SCN_H3_NO_PE
constexpr bool kNoPe = true;
#else
constexpr bool kNoPe = false;
#endif
#ifdef SCN_H3_NO_STREAM
constexpr bool kNoStream = true;
#else
constexpr bool kNoStream = false;
#endif
#ifdef SCN_H3_NO_BARRIER
constexpr bool kNoBarrier = true;
#else
constexpr bool kNoBarrier = false;
#endif
#ifdef SCN_H3_LATE_LOADS
constexpr bool kLateLoads = true;
#else
constexpr bool kLateLoads = false;
#endif
#ifdef SCN_PLAIN_STORE
constexpr bool kPlainStore = true;
#else
constexpr bool kPlainStore = false;
#endif
#ifdef SCN_BURST
constexpr bool kBurst = true;
#else
constexpr bool kBurst = false;
#endif
#ifdef SCN_GROUP_HINTS
constexpr bool kGroupHints = true;
#else
constexpr bool kGroupHints = false;
#endif
#ifdef SCN_NO_CHAIN
constexpr bool kNoChain = true;
#else
constexpr bool kNoChain = false;
#endif
}  // namespace lab
}  // namespace scn

// scn_lab.h -- switches of the timing experiments on the resident kernels (tools/ablate_h3.sh).  The PRODUCT's copy:
// every switch is a compile-time `false`, the guarded statements fold away.  The experiments are built from the same
// kernel sources with tools/lab/scn_lab.h in front of this one on the include path, which sets the switches from -D flags;
// nothing in the product's sources tests a macro.
#pragma once
namespace scn {
namespace lab {
constexpr bool kNoEpilogue = false;      // epilogue reduced to moving the accumulators into the operand planes
constexpr bool kNoStore = false;         // no activation / gradient stores
constexpr bool kNoPeStore = false;       // no stores of the encodings
constexpr bool kNoPe = false;            // no sines / cosines
constexpr bool kNoStream = false;        // no weight stream (LDS keeps its first chunks)
constexpr bool kNoBarrier = false;       // no chunk barrier (racy)
constexpr bool kLateLoads = false;       // the stream's global loads in slots 24 .. 47 instead of 0 .. 23
// the fp32-MFMA yardstick kernels (mlp_common.h)
constexpr bool kPlainStore = false;      // workspace stores with the default cache policy instead of non-temporal
constexpr bool kBurst = false;           // round 1's burst schedule of a chunk's memory instructions instead of the spread one
constexpr bool kGroupHints = false;      // explicit MFMA / VALU interleave hints in the last chunk
constexpr bool kNoChain = false;         // no accumulator chaining between chunks
}  // namespace lab
}  // namespace scn

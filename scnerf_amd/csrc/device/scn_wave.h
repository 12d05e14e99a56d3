// scn_wave.h -- the small CDNA4 (gfx950, wave64) vocabulary the kernels are written in.
//
// Everything cross-lane or matrix-core goes through these wrappers so that the kernel
// sources read as plain HIP.  This is the device implementation (the only one shipped).
// tests/emu/shim/ holds a same-named header that the CPU SIMT interpreter used by the
// `-m "not gpu"` logic tests puts first on the include path; the product never sees it.
#pragma once
#include <hip/hip_runtime.h>

namespace scn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// D[32x32] += A[32x2] * B[2x32], exact fp32 (a k-ordered fmaf chain), 64 cycles/SIMD.
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; it receives column
// j = l&31 of D, rows (r&3) + 8*(r>>2) + 4*(l>>5) for r = 0..15.
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// D[32x32] += A[32x16] * B[16x32], bf16 operands (raw bits in shorts), fp32 accumulate, 32 cycles/SIMD:
// 16x the rate of the exact-fp32 form above.  Lane l supplies A[i = l&31][k = 8 (l>>5) .. +7] and
// B[k = 8 (l>>5) .. +7][j = l&31]; C/D layout as mfma_32x32x2.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_f16: the same shape on fp16 operands (bit patterns in shorts)
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// fp32 -> fp16 (round to nearest even) as a bit pattern in the low half of a word, and back
__device__ __forceinline__ unsigned f16_bits(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float f16_value(unsigned bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits); }
// the low halves of two words as one word (lo_word's in the low half)
__device__ __forceinline__ unsigned low_halves(unsigned lo_word, unsigned hi_word) {
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x05040100u);
}
// ---- two-way fp16 cut of an fp32 value x scaled by a power of two s:  x s = h + l + O(2^-22 |x s|) ----------------
// {f16(a s), f16(b s)} in the low / high half of one word (v_fma_mixlo_f16 / v_fma_mixhi_f16: the product is exact,
// one rounding to nearest even)
__device__ __forceinline__ unsigned pack_f16_scaled(float a, float b, float s) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(d) : "v"(a), "v"(b), "v"(s));
    return d;
}
// a s - (half HI of `packed`), exact (v_fma_mix_f32 with an fp16 third operand): what the low plane has to carry
template <int HI>
__device__ __forceinline__ float residual_f16(float a, float s, unsigned packed) {
    float r;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(packed));
    else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(packed));
    return r;
}
// {f16(a), f16(b)}, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    unsigned d;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// max(m, a, b) and max(m, |a|, |b|) as one v_max3_f32
__device__ __forceinline__ float max3(float m, float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max3_abs(float m, float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
// ds_read_b64_tr_b16: every lane names 8 bytes of LDS (4 x 16 bit, 8-byte aligned); the 16 lanes of a group
// together name a [4 rows][16 columns] block (lane c: row c >> 2, columns 4 (c & 3) .. +3) and lane c
// receives column c, rows 0 .. 3 -- the transpose an MFMA operand needs when the LDS image is k-major.
__device__ __forceinline__ s16x4 lds_read_tr16(const short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
// {bits 31..16 of lo_word, bits 31..16 of hi_word} as one dword (v_perm_b32): two truncated bf16
__device__ __forceinline__ unsigned high_halves(unsigned lo_word, unsigned hi_word) {
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}
// A global pointer known to be the same in every lane, made OPAQUE to the compiler (an empty asm on its two SGPR halves): base of
// `global_load v, v_off, s[base]`.  Without it the compiler folds loop-invariant per-lane offsets into 64-bit VGPR
// pointers (2 VGPRs per distinct offset, hoisted out of the loop, spilled).  The pointer stays in the global address
// space (rebuilt from integers as a generic pointer it would produce flat loads).
typedef const __attribute__((address_space(1))) char* global_bytes;
__device__ __forceinline__ global_bytes uniform_global_bits(unsigned long long v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm("" : "+s"(lo), "+s"(hi));            // (readfirstlane of an already-scalar value folds away; this does not)
    return reinterpret_cast<global_bytes>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ global_bytes uniform_global(const void* p) {
    return uniform_global_bits(reinterpret_cast<unsigned long long>(p));
}
__device__ __forceinline__ global_bytes uniform_global(global_bytes p) {
    return uniform_global_bits(reinterpret_cast<unsigned long long>(p));
}
// clocks for lab measurements: shader-core cycles (follow the DVFS clock) and the constant 100 MHz reference
__device__ __forceinline__ unsigned long long shader_cycles() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned long long reference_ticks() { return __builtin_amdgcn_s_memrealtime(); }
// A per-lane value the compiler must recompute where it is used: keeps address arithmetic that is invariant in an
// outer loop from being hoisted into long-lived registers (which then spill around a 256-AGPR MFMA loop).
__device__ __forceinline__ unsigned pinned_here(unsigned x) {
    asm volatile("" : "+v"(x));
    return x;
}
// The kernel's own argument block, through a pointer the optimiser cannot connect with the arguments it has already loaded:
// a field read through it is fetched (s_load) where it is used.  For arguments needed only at the very end of a long kernel
// -- the fine stage's compositing outputs -- this keeps two dozen SGPRs from living across the whole body.  `mirror`: the
// parameter list restated as a struct (the argument block IS that struct: same members, same order, natural alignment);
// ignored here, it is what the CPU interpreter's twin of this function returns the address of.
template <class Args>
__device__ __forceinline__ const __attribute__((address_space(4))) Args* late_args(const Args& mirror) {
    (void)mirror;
    const unsigned long long v = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const __attribute__((address_space(4))) Args*>(((unsigned long long)hi << 32) | lo);
}
// A wave-uniform integer the optimiser must take as NEW where this is called (a volatile empty asm on its SGPR halves):
// address arithmetic built on it is not invariant in an enclosing loop, so it is not hoisted into registers that then live
// -- and spill -- across the whole loop body (the fine stage's loop over a workgroup's wave tiles).
__device__ __forceinline__ long fresh_uniform(long v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long)v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ int fresh_uniform(int v) {
    int x = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ f32x4 load_f32x4(global_bytes base, unsigned lane_off) {
    return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + lane_off);
}
// the same for data read once (does not displace what the caches hold)
__device__ __forceinline__ f32x4 load_stream_f32x4(global_bytes base, unsigned lane_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + lane_off));
}
// stores through a wave-uniform base + 32-bit lane offset (`global_store v_off, v[data], s[base]`)
typedef __attribute__((address_space(1))) char* global_bytes_rw;
__device__ __forceinline__ global_bytes_rw uniform_global_rw(void* p) {
    return const_cast<global_bytes_rw>(uniform_global(p));
}
__device__ __forceinline__ global_bytes_rw uniform_global_rw(global_bytes_rw p) {
    return const_cast<global_bytes_rw>(uniform_global(static_cast<global_bytes>(p)));
}
template <typename V>
__device__ __forceinline__ void store_at(global_bytes_rw base, unsigned lane_off, V v) {
    *reinterpret_cast<__attribute__((address_space(1))) V*>(base + lane_off) = v;
}
template <typename V>
__device__ __forceinline__ void store_stream_at(global_bytes_rw base, unsigned lane_off, V v) {
    __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((address_space(1))) V*>(base + lane_off));
}
// The activation / gradient workspaces of the resident kernels: 16 bytes per lane, written once, read once by a LATER launch (the
// weight-gradient GEMMs).  Stored with system scope + non-temporal (`sc0 sc1 nt`: written through, nothing left behind in the
// cache hierarchy of a socket that is power-bound while these kernels run): 3 % less time per eight-layer chain than `nt` alone
// in the lab, 1.5 % per training step in the product on one box (profiles/r06_lab_store_policy.txt).  Visibility to the next
// launch is the kernel boundary's, as before (the whole GPU suite passes, incl. the bit-reproducibility tests).
// Emitted as a raw BUFFER store: the global-store builtins cannot carry this policy, the buffer builtin takes it as its cache-policy
// immediate (sc0 = 1, nt = 2, sc1 = 16), and -- unlike an inline asm, the first form of this function -- the compiler sees the
// instruction: it keeps the hardware's "store of more than 8 bytes, then a write of its data registers" hazard and the wait for
// pending loads into `v` itself.  (The asm form needed an `s_nop 1` behind every store -- without it one stored value in a
// thousand was the NEXT value written to that register, found by the parity tests -- and could not take LDS-sourced data.)
// The descriptor: base = the wave-uniform address, stride 0, 2 GB window, the raw 32-bit format word of gfx9.
typedef unsigned scn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_written_through_at(global_bytes_rw base, unsigned lane_off, f32x4 v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(scn_u32x4, v), rsrc, (int)lane_off, 0, 1 | 2 | 16);
}
template <typename V>
__device__ __forceinline__ V load_at(global_bytes base, unsigned lane_off) {
    return *reinterpret_cast<const __attribute__((address_space(1))) V*>(base + lane_off);
}
// 16-byte global load that does not displace what the caches hold (streamed-once operands)
__device__ __forceinline__ f32x4 load_stream(const f32x4* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ float shfl(float v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float shfl_up(float v, int delta) { return __shfl_up(v, delta, 64); }
__device__ __forceinline__ float shfl_down(float v, int delta) { return __shfl_down(v, delta, 64); }
__device__ __forceinline__ double shfl(double v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double shfl_up(double v, int delta) { return __shfl_up(v, delta, 64); }
__device__ __forceinline__ double shfl_down(double v, int delta) { return __shfl_down(v, delta, 64); }
__device__ __forceinline__ int shfl(int v, int src_lane) { return __shfl(v, src_lane, 64); }
__device__ __forceinline__ int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int shfl_up(int v, int delta) { return __shfl_up(v, delta, 64); }

// the value lane `k` holds, k the same in every lane (a wave-uniform index: v_readlane_b32, no LDS crossbar, no latency
// to speak of -- what the samplers' serial sections are built on)
__device__ __forceinline__ float read_lane(float v, int k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
__device__ __forceinline__ double read_lane(double v, int k) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popcount64(unsigned long long m) { return __popcll(m); }

__device__ __forceinline__ void block_sync() { __syncthreads(); }
// The lanes of ONE wave hand data to each other through LDS: a wave's LDS instructions execute in order, so nothing is
// emitted -- the call pins the order for the compiler (and is the rendezvous of the wave's fibers on the CPU interpreter).
__device__ __forceinline__ void wave_lds_handoff() { __builtin_amdgcn_wave_barrier(); }

// The kernel claims the SIMD's whole register file (256 architectural + 256 accumulation registers): no instruction is emitted, the
// clobbers make v255 and a255 part of the kernel's allocation.  For the kernels built around ONE wave per SIMD on the fp16 matrix
// pipe (csrc/mlp_h3.h, wgrad256_half.h, wgrad_half_narrow.h): with 425 .. 444 registers they left room for a 64-register wave of
// ANOTHER kernel on their SIMD -- never this library's own (one stream), but a second process's or another stream's -- and such a
// guest lost writes to lanes 48..63 of a register while the host wave ran its MFMA chain (tools/guest_probe.py,
// profiles/r05_two_process_probe.txt: the camera backward kernel beside the data-gradient kernel, 7 % of its launches).  A full
// allocation admits no guest; for a kernel that runs one wave per SIMD anyway it costs nothing.  A MITIGATION, not a root cause: a
// synthetic host of the same shape does not reproduce the loss (tools/ubench/guest_write_lab.hip, profiles/r06_guest_write_lab.txt:
// 0 of ~112 000 guest launches), so what in these kernels' instruction streams does it is open.
__device__ __forceinline__ void claim_whole_register_file() { asm volatile("" ::: "v255", "a255"); }

// Compiler-only fence: instructions are not moved across it by the machine scheduler (used to keep
// prefetches where they were written; no instruction is emitted).
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// scheduling-group hints (LLVM AMDGPU sched_group_barrier): the next `n` MFMA / VALU instructions of the
// region form one group; groups are emitted in the order the hints appear
// tells the compiler a value is the same in every lane of the wave (lives in an SGPR; branches on it are real)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// max without the canonicalising self-max the compiler adds in front of fmaxf under IEEE mode (v_max_f32
// quiets signalling NaNs by itself): one instruction, and VALU instructions are not free next to fp32
// MFMAs -- both run on the SIMD's fp32 lanes (tools/ubench/mfma_valu.hip: ~4.5 cycles per VALU op).
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(x, 0) with the zero as an inline constant (max_raw(x, 0.f) would park the constant in a register)
__device__ __forceinline__ float relu_raw(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// a + b as ONE v_add_f32 (plain `+` on neighbouring values is SLP-packed into v_pk_add_f32, which is slower
// than two scalar adds next to fp32 MFMAs: MI355X_MICROARCH.md, "price of one filler beside MFMAs")
__device__ __forceinline__ float add_raw(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// bits = 2 * bits + (v > 0): one compare into VCC and one add-with-carry
__device__ __forceinline__ unsigned shift_in_positive(unsigned bits, float v) {
    asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(v) : "vcc");
    return bits;
}
// v if bit `pos` of `bits` is set, else +0: sign-extended 1-bit field extract (0 / ~0) and one AND
template <int POS>
__device__ __forceinline__ float keep_if_bit(float v, unsigned bits) {
    int m;
    float r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(POS));
    asm("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}
template <int N> __device__ __forceinline__ void sched_group_mfma() { __builtin_amdgcn_sched_group_barrier(0x008, N, 0); }
template <int N> __device__ __forceinline__ void sched_group_valu() { __builtin_amdgcn_sched_group_barrier(0x002, N, 0); }

// Dynamic LDS of the launch (16-byte aligned; no static __shared__ anywhere, so the
// dynamic region starts at offset 0: cdna_hip_programming.md guideline 17).
template <typename T>
__device__ __forceinline__ T* dynamic_lds() {
    extern __shared__ __attribute__((aligned(16))) char scn_lds_raw[];
    return reinterpret_cast<T*>(scn_lds_raw);
}

__device__ __forceinline__ float atomic_add(float* p, float v) { return atomicAdd(p, v); }
// *p = max(*p, v) for NON-NEGATIVE floats (their bit patterns order like unsigned integers); no return value needed
__device__ __forceinline__ void atomic_max_nonneg(float* p, float v) {
    atomicMax(reinterpret_cast<unsigned*>(p), __float_as_uint(v));
}

__device__ __forceinline__ void sincos(float x, float* s, float* c) { sincosf(x, s, c); }

}  // namespace scn

// TEST INFRASTRUCTURE -- CPU SIMT-interpreter implementation of the scn:: wave vocabulary
// (same names and semantics as scnerf_amd/csrc/device/scn_wave.h; put FIRST on the include
// path by tests/emu/build_emu.py, never by the product build).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace scn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

inline int lane_id() { return simt::cur_lane(); }
inline int wave_id() { return (int)(threadIdx.x >> 6); }

// v_mfma_f32_32x32x2_f32: lane l gives A[l&31][l>>5], B[l>>5][l&31]; receives column l&31,
// rows (r&3)+8*(r>>2)+4*(l>>5); D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) (k-ordered chain).
inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    uint64_t pack;
    uint32_t ua, ub;
    std::memcpy(&ua, &a, 4);
    std::memcpy(&ub, &b, 4);
    pack = (uint64_t)ua | ((uint64_t)ub << 32);
    const uint64_t* x = simt::wave_exchange(pack);
    const int l = simt::cur_lane();
    const int j = l & 31, hi = l >> 5;
    auto A = [&](int i, int k) { uint32_t u = (uint32_t)(x[i + 32 * k] & 0xffffffffu); float f; std::memcpy(&f, &u, 4); return f; };
    auto B = [&](int k, int jj) { uint32_t u = (uint32_t)(x[jj + 32 * k] >> 32); float f; std::memcpy(&f, &u, 4); return f; };
    const float b0 = B(0, j), b1 = B(1, j);
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        c[r] = std::fmaf(A(i, 1), b1, std::fmaf(A(i, 0), b0, c[r]));
    }
    return c;
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
inline float bf16_bits_to_float(short h) {
    const uint32_t u = (uint32_t)(uint16_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// v_mfma_f32_32x32x16_bf16: lane l gives A[l&31][8 (l>>5) .. +7], B[8 (l>>5) .. +7][l&31]; the sixteen
// products (exact in fp32) are summed here in double and added to C with one rounding.
inline f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
    uint64_t mine[4];
    std::memcpy(&mine[0], &a, 16);
    std::memcpy(&mine[2], &b, 16);
    uint64_t all[4][64];
    for (int w = 0; w < 4; ++w) {
        const uint64_t* x = simt::wave_exchange(mine[w]);
        std::memcpy(all[w], x, sizeof(all[w]));
    }
    auto elem = [&](int word0, int lane, int k) {          // element k (0..7) of lane's operand
        const uint64_t u = all[word0 + (k >> 2)][lane];
        return bf16_bits_to_float((short)(uint16_t)(u >> (16 * (k & 3))));
    };
    const int l = simt::cur_lane();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double sum = 0.0;
        for (int k = 0; k < 16; ++k) sum += (double)elem(0, i + 32 * (k >> 3), k & 7) * (double)elem(2, j + 32 * (k >> 3), k & 7);
        c[r] = (float)((double)c[r] + sum);
    }
    return c;
}
inline unsigned f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}
inline float f16_value(unsigned bits) {
    const unsigned short u = (unsigned short)bits;
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
}
inline unsigned low_halves(unsigned lo_word, unsigned hi_word) { return (lo_word & 0xffffu) | (hi_word << 16); }
inline unsigned pack_f16_scaled(float a, float b, float s) { return (f16_bits(a * s) & 0xffffu) | (f16_bits(b * s) << 16); }
template <int HI>
inline float residual_f16(float a, float s, unsigned packed) { return std::fmaf(a, s, -f16_value(HI ? packed >> 16 : packed & 0xffffu)); }
inline unsigned pack_f16(float a, float b) { return (f16_bits(a) & 0xffffu) | (f16_bits(b) << 16); }
inline float max3(float m, float a, float b) { return std::fmax(m, std::fmax(a, b)); }
inline float max3_abs(float m, float a, float b) { return std::fmax(m, std::fmax(std::fabs(a), std::fabs(b))); }
inline f32x16 mfma_32x32x16_f16(s16x8 a, s16x8 b, f32x16 c) {
    uint64_t mine[4];
    std::memcpy(&mine[0], &a, 16);
    std::memcpy(&mine[2], &b, 16);
    uint64_t all[4][64];
    for (int w = 0; w < 4; ++w) {
        const uint64_t* x = simt::wave_exchange(mine[w]);
        std::memcpy(all[w], x, sizeof(all[w]));
    }
    auto elem = [&](int word0, int lane, int k) {
        const uint64_t u = all[word0 + (k >> 2)][lane];
        return f16_value((unsigned)(uint16_t)(u >> (16 * (k & 3))));
    };
    const int l = simt::cur_lane();
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double sum = 0.0;
        for (int k = 0; k < 16; ++k) sum += (double)elem(0, i + 32 * (k >> 3), k & 7) * (double)elem(2, j + 32 * (k >> 3), k & 7);
        c[r] = (float)((double)c[r] + sum);
    }
    return c;
}
// ds_read_b64_tr_b16: lane c of a 16-lane group names row c >> 2, columns 4 (c & 3) .. +3 of the group's
// [4][16] block and receives column c
inline s16x4 lds_read_tr16(const short* p) {
    uint64_t mine;
    std::memcpy(&mine, p, 8);
    const uint64_t* x = simt::wave_exchange(mine);
    const int l = simt::cur_lane(), base = l & ~15, c = l & 15;
    s16x4 out;
    for (int row = 0; row < 4; ++row) out[row] = (short)(uint16_t)(x[base + 4 * row + (c >> 2)] >> (16 * (c & 3)));
    return out;
}
inline unsigned high_halves(unsigned lo_word, unsigned hi_word) { return (lo_word >> 16) | (hi_word & 0xffff0000u); }
inline f32x4 load_stream(const f32x4* p) { return *p; }
typedef const char* global_bytes;
inline global_bytes uniform_global(const void* p) { return static_cast<const char*>(p); }
inline global_bytes uniform_global(global_bytes p) { return p; }
inline unsigned pinned_here(unsigned x) { return x; }
template <class Args> inline const Args* late_args(const Args& mirror) { return &mirror; }
inline long fresh_uniform(long v) { return v; }
inline int fresh_uniform(int v) { return v; }
inline unsigned long long shader_cycles() { return 0; }
inline unsigned long long reference_ticks() { return 0; }
typedef char* global_bytes_rw;
inline global_bytes_rw uniform_global_rw(void* p) { return static_cast<char*>(p); }
inline global_bytes_rw uniform_global_rw(global_bytes_rw p) { return p; }
template <typename V>
inline void store_at(global_bytes_rw base, unsigned lane_off, V v) { std::memcpy(base + lane_off, &v, sizeof(V)); }
template <typename V>
inline void store_stream_at(global_bytes_rw base, unsigned lane_off, V v) { std::memcpy(base + lane_off, &v, sizeof(V)); }
inline void store_written_through_at(global_bytes_rw base, unsigned lane_off, f32x4 v) { std::memcpy(base + lane_off, &v, sizeof(v)); }
template <typename V>
inline V load_at(global_bytes base, unsigned lane_off) {
    V v;
    std::memcpy(&v, base + lane_off, sizeof(V));
    return v;
}
inline f32x4 load_stream_f32x4(global_bytes base, unsigned lane_off) {
    f32x4 v;
    std::memcpy(&v, base + lane_off, 16);
    return v;
}
inline f32x4 load_f32x4(global_bytes base, unsigned lane_off) {
    f32x4 v;
    std::memcpy(&v, base + lane_off, 16);
    return v;
}

template <typename T>
inline T exchange_read(T v, int src) {
    static_assert(sizeof(T) <= 8, "");
    uint64_t u = 0;
    std::memcpy(&u, &v, sizeof(T));
    const uint64_t* x = simt::wave_exchange(u);
    T out;
    const uint64_t s = x[src & 63];
    std::memcpy(&out, &s, sizeof(T));
    return out;
}
template <typename T> inline T shfl_t(T v, int src) { return exchange_read(v, src); }
template <typename T> inline T shfl_xor_t(T v, int m) { return exchange_read(v, simt::cur_lane() ^ m); }
template <typename T> inline T shfl_up_t(T v, int d) { int l = simt::cur_lane(); return exchange_read(v, l - d >= 0 ? l - d : l); }
template <typename T> inline T shfl_down_t(T v, int d) { int l = simt::cur_lane(); return exchange_read(v, l + d < 64 ? l + d : l); }

inline float shfl(float v, int s) { return shfl_t(v, s); }
inline float shfl_xor(float v, int m) { return shfl_xor_t(v, m); }
inline float shfl_up(float v, int d) { return shfl_up_t(v, d); }
inline float shfl_down(float v, int d) { return shfl_down_t(v, d); }
inline double shfl(double v, int s) { return shfl_t(v, s); }
inline double shfl_xor(double v, int m) { return shfl_xor_t(v, m); }
inline double shfl_up(double v, int d) { return shfl_up_t(v, d); }
inline double shfl_down(double v, int d) { return shfl_down_t(v, d); }
inline int shfl(int v, int s) { return shfl_t(v, s); }
inline float read_lane(float v, int k) { return shfl_t(v, k); }
inline double read_lane(double v, int k) { return shfl_t(v, k); }
inline int shfl_xor(int v, int m) { return shfl_xor_t(v, m); }
inline int shfl_up(int v, int d) { return shfl_up_t(v, d); }

inline unsigned long long ballot(bool p) {
    const uint64_t* x = simt::wave_exchange(p ? 1u : 0u);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(x[i] & 1u) << i;
    return m;
}
inline int popcount64(unsigned long long m) { return __builtin_popcountll(m); }

inline void block_sync() { simt::block_barrier(); }
inline void wave_lds_handoff() { (void)simt::wave_exchange(0); }       // every fiber of the wave has made its LDS writes
inline void sched_fence() {}
inline void claim_whole_register_file() {}
inline int uniform(int v) { return v; }
template <typename T> inline const T* uniform_ptr(const T* p) { return p; }
inline float max_raw(float a, float b) { return a > b ? a : b; }
inline float add_raw(float a, float b) { return a + b; }
inline float relu_raw(float x) { return x > 0.f ? x : 0.f; }
template <int POS> inline float keep_if_bit(float v, unsigned bits) { return ((bits >> POS) & 1u) ? v : 0.f; }
inline unsigned shift_in_positive(unsigned bits, float v) { return bits + bits + (v > 0.f ? 1u : 0u); }
template <int N> inline void sched_group_mfma() {}
template <int N> inline void sched_group_valu() {}

template <typename T>
inline T* dynamic_lds() { return reinterpret_cast<T*>(simt::block_lds()); }

inline float atomic_add(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        std::memcpy(&f, &old, 4);
        const float nf = f + v;
        uint32_t nu;
        std::memcpy(&nu, &nf, 4);
        if (__atomic_compare_exchange_n(u, &old, nu, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}

inline void atomic_max_nonneg(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t nv;
    std::memcpy(&nv, &v, 4);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED);
    while (old < nv && !__atomic_compare_exchange_n(u, &old, nv, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

inline void sincos(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }

}  // namespace scn

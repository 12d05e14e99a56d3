// mlp_fwd_h3_kernel.h -- the resident-arithmetic forward kernel (see mlp_fwd_h3.hip for what it replaces) and its launch
// templates.  Included by one translation unit per instantiation group (mlp_fwd_h3_pd3.hip, _pd4.hip, _coarse.hip): the
// kernel is a long straight-line program, and compiled side by side its variants take a third of the wall time.
#pragma once
#include <scn_wave.h>

#include "launch.h"
#include "mlp_fwd_h3_api.h"
#include "mlp_h3.h"
#include "ray_sample.h"
#include "ray_stage.h"

namespace scn {
namespace h3f {

using namespace scn::mlp;
using namespace scn::h3;


// Encodings are saved in torch column order so the wgrad GEMM writes weight columns directly (as mlp_fwd.hip), row-major
// [P][LD] -- as COALESCED stores.  One float per lane and instruction into 32 different rows (what mlp_fwd.hip's
// store_pe does) was a cache line per lane: 0.46 of the fine forward's 5.8 Mcycles (tools/ablate_h3.sh nopestore,
// profiles/r05_ablation_h3.txt).
// A wave tile's 32 rows of a row-major section are ONE contiguous block of 32 x LD floats, and the slots are (or can be put)
// in LDS in slot layout -- `slots[g * kThreads + thread]` = slots 4 g .. 4 g + 3 of the thread --, so the block goes out as
// LD / 8 instructions of 16 bytes per lane: lane l of instruction k owns columns 4 (l % (LD / 4)) .. + 3 of row
// k * (256 / LD) + l / (LD / 4) and gathers them from the slot layout (column c = slot s_c of lane half h_c: PeInverse).
template <int PD, int L, int NS, int LD>
struct PeInverse {
    unsigned off[LD];          // byte offset of column c's slot relative to its sample's lane-half-0 entry; ~0u: zero pad
    constexpr PeInverse() : off() {
        for (int c = 0; c < LD; ++c) off[c] = 0xffffffffu;
        for (int s = 0; s < NS; ++s)
            for (int h = 0; h < 2; ++h) {
                const int c = pe_col(L, s, h, PD);
                if (c >= 0) off[c] = (unsigned)((s / 4) * (kThreads * 16) + h * 512 + (s % 4) * 4);
            }
    }
};
template <int PD, int L, int NS, int LD>
__device__ __forceinline__ void store_pe_tile(const char* slots, float* __restrict__ section, long wave_tile, int wave, int lane) {
    if (lab::kNoPeStore) return;
    wave_lds_handoff();                               // the slots are the wave's own lanes' writes
    static constexpr PeInverse<PD, L, NS, LD> inverse{};
    constexpr int LPR = LD / 4;                       // lanes per row
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = inverse.off[4 * (lane % LPR) + i];
    const global_bytes_rw block = uniform_global_rw(section + wave_tile * (32L * LD));
#pragma unroll
    for (int k = 0; k < LD / 8; ++k) {
        const int m = k * (64 / LPR) + lane / LPR;
        const char* row = slots + (wave * 64 + m) * 16;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = o[i] == 0xffffffffu ? 0.f : *reinterpret_cast<const float*>(row + o[i]);
        store_stream_at(uniform_global_rw(block + k * 1024), pinned_here((unsigned)lane * 16u), v);
    }
}

// KIND 0: ReLU + mask bits;  1: the same + the density head's dot product (layer 7);  2: linear (feature layer)
// members only some kinds use
struct NoDensity {};
struct Density {
    float sg;                  // running dot product of the density head (this lane's half of the features)
    const float* alpha;        // its weights, LDS lane-vector table + 4 h
    f32x4 wq;
};
template <bool ON> struct MaskBits { unsigned bits0, bits1, words[4]; };
template <> struct MaskBits<false> {};

template <bool TRAIN, int KIND>
struct FwdEpi : std::conditional_t<KIND == 1, Density, NoDensity>, MaskBits<TRAIN && KIND != 2> {
    float os, s_next, am;
    const float* bias;         // LDS lane-vector table of the layer, + 4 h
    global_bytes_rw save;      // this wave tile's block of the layer's section (TRAIN)
    unsigned lane16;
    // (reading the bias piece one piece ahead into a second register was tried: no difference, four registers more)
    f32x4 bq;
    float v[4];
    unsigned hp;

    __device__ __forceinline__ void prime() {}

    template <int P, int PIECE, int SUB, int NS>
    __device__ __forceinline__ void sub(f32x16 (&acc)[2], u32x4 (&oh)[NS], u32x4 (&ol)[NS]) {
        constexpr int x = PIECE >> 2, q = PIECE & 3, T = 2 * P + x;
        constexpr int sl = 2 * T + (q >> 1), c0 = 2 * (q & 1);
        static_assert(sl < NS, "operand buffer too small for this tile");
        if constexpr (lab::kNoEpilogue) {
            if constexpr (SUB == 1) {
                oh[sl][c0] = __float_as_uint(acc[x][4 * q]) & 0x3bff3bffu; oh[sl][c0 + 1] = __float_as_uint(acc[x][4 * q + 1]) & 0x3bff3bffu;
                ol[sl][c0] = __float_as_uint(acc[x][4 * q + 2]) & 0x3bff3bffu; ol[sl][c0 + 1] = __float_as_uint(acc[x][4 * q + 3]) & 0x3bff3bffu;
            }
        } else if constexpr (SUB == 0) {
            bq = *reinterpret_cast<const f32x4*>(bias + (4 * T + q) * 8);
            if constexpr (KIND == 1) this->wq = *reinterpret_cast<const f32x4*>(this->alpha + (4 * T + q) * 8);
            if constexpr (TRAIN && KIND != 2 && PIECE == 0) { this->bits0 = 0u; this->bits1 = 0u; }
        } else if constexpr (SUB <= 4) {
            constexpr int e = SUB - 1;
            const float z = __builtin_fmaf(acc[x][4 * q + e], os, bq[e]);
            v[e] = KIND == 2 ? z : relu_raw(z);
            if constexpr (TRAIN && KIND != 2) {
                if constexpr (x == 0) this->bits0 = shift_in_positive(this->bits0, v[e]);
                else this->bits1 = shift_in_positive(this->bits1, v[e]);
            }
        } else if constexpr (SUB == 5) {
            if constexpr (KIND == 2) { am = max3_abs(am, v[0], v[1]); am = max3_abs(am, v[2], v[3]); }
            else { am = max3(am, v[0], v[1]); am = max3(am, v[2], v[3]); }
        } else if constexpr (SUB == 6) {
            hp = pack_f16_scaled(v[0], v[1], s_next);
        } else if constexpr (SUB == 7) {
            oh[sl][c0] = hp;
            ol[sl][c0] = pack_f16(residual_f16<0>(v[0], s_next, hp), residual_f16<1>(v[1], s_next, hp));
        } else if constexpr (SUB == 8) {
            hp = pack_f16_scaled(v[2], v[3], s_next);
        } else if constexpr (SUB == 9) {
            oh[sl][c0 + 1] = hp;
            ol[sl][c0 + 1] = pack_f16(residual_f16<0>(v[2], s_next, hp), residual_f16<1>(v[3], s_next, hp));
        } else if constexpr (SUB == 10) {
            if constexpr (TRAIN && !lab::kNoStore) {
                // (wave-uniform base + the lane's 32-bit offset: as 64-bit per-lane pointers the eight piece bases of a
                //  layer are hoisted into sixteen long-lived registers)
                store_written_through_at(uniform_global_rw(save + (4 * T + q) * 1024), pinned_here(lane16), f32x4{v[0], v[1], v[2], v[3]});
            }
        } else {
            if constexpr (KIND == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) this->sg = __builtin_fmaf(this->wq[e], v[e], this->sg);
            }
            if constexpr (TRAIN && KIND != 2 && PIECE == 7) this->words[P] = (this->bits0 << 16) | this->bits1;
        }
    }
};



// offsets (floats) of the lane-vector tables inside the LDS copy of wpk[kFwdBias ..)
constexpr int kTabFeat = 8 * 256, kTabViews = kTabFeat + 256, kTabRgb = kTabViews + 128, kTabAlpha = kTabRgb + 32,
              kTabAlphaB = kTabAlpha + 256;

// STAGE: what the launch does around the network
//   kPoints  the points are given (`pts`), one wave tile of 32 samples per wave, raw out
//   kCoarse  the whole coarse stage (NeRF/render.py:235-262): stratified depths in front, compositing behind; a workgroup =
//            2 rays x 64 samples
//   kFine    the whole fine stage (NeRF/render.py:269-285): the inverse-CDF sampler + merge in front (ray_sample.h, the
//            stand-alone sampler's own instructions), compositing behind.  A ray has 64 + N_importance samples, a workgroup
//            128 per pass over the weight stream: it takes `fs.rays_per_block` whole rays (1, 2, 1 for 128, 192, 256
//            samples per ray) through `fs.tiles` passes (1, 3, 2), the depths and the raw outputs of its rays parked in LDS.
enum : int { kPoints = 0, kCoarse = 1, kFine = 2 };
constexpr int kFineMaxSamples = 384;               // samples of a fine-stage workgroup (rays_per_block x samples per ray)

template <int PD, int STAGE = kPoints>
__host__ __device__ constexpr unsigned fwd_lds_bytes() {
    return (unsigned)(kStreamLds + kTableFloats * 4 + Var<PD>::kES * kThreads * 4 + (STAGE == kFine ? kFineMaxSamples * 20 : 0));
}

// the parameter list of mlp_fwd_h3_kernel as its argument block lays it out (late_args)
struct FwdKernelArgs {
    const float* pts; const float* viewdirs; int vd_stride; int samples_per_ray; const float* wpk; const short* wh3;
    const float* sc; float* raw; float* save; long P; CoarseStage cs; FineStage fs; ChunkMaxima cm;
};
static_assert(sizeof(FwdKernelArgs) == 72 + sizeof(CoarseStage) + sizeof(FineStage) + sizeof(ChunkMaxima), "argument block layout");

template <int PD, bool TRAIN, int STAGE>
__global__ __launch_bounds__(kThreads, 1) void mlp_fwd_h3_kernel(
    const float* __restrict__ pts, const float* __restrict__ viewdirs, int vd_stride, int samples_per_ray,
    const float* __restrict__ wpk, const short* __restrict__ wh3, const float* __restrict__ sc,
    float* __restrict__ raw, float* __restrict__ save_arg, long P, CoarseStage cs, FineStage fs, ChunkMaxima cm) {
    using V = Var<PD>;
    constexpr int ES = V::kES, NE = ES / 8;
    constexpr bool COARSE = STAGE == kCoarse, FINE = STAGE == kFine;
    claim_whole_register_file();
    float* const save = TRAIN ? save_arg : nullptr;
    const long Ppad_kernel = padded_samples(P);

    Wave w;
    w.lds = dynamic_lds<char>();
    float* const tables = reinterpret_cast<float*>(w.lds + kStreamLds);
    // fine stage: the merged depths and the raw outputs of the workgroup's rays (behind the parked encoding)
    float* const zbuf = reinterpret_cast<float*>(w.lds + kStreamLds + kTableFloats * 4 + ES * kThreads * 4);      // [384]
    f32x4* const rawbuf = reinterpret_cast<f32x4*>(zbuf + kFineMaxSamples);                                        // [384]
    for (int i = threadIdx.x; i < V::kFwdTotal - V::kFwdBias; i += kThreads) tables[i] = wpk[V::kFwdBias + i];
    const int tiles = FINE ? uniform(fs.tiles) : 1;
    if constexpr (FINE) {
        static_assert(PD == 3, "the fine stage samples 3-D points");
        // every wave runs the sampler (its pieces synchronise the workgroup): waves 0 .. rays_per_block - 1 own the
        // workgroup's rays, the others repeat one of them into scratch of their own and write nothing
        const int wv = uniform(wave_id());
        const int slot = wv % fs.rays_per_block;
        long ray = (long)blockIdx.x * fs.rays_per_block + slot;
        const bool mine = wv < fs.rays_per_block && ray < fs.n_rays;
        if (ray >= fs.n_rays) ray = fs.n_rays - 1;
        const int tot = kCoarseSamples + fs.n_importance;
        float* scratch = reinterpret_cast<float*>(w.lds + kStreamLds + kTableFloats * 4) + wv * ray::fine_sample_lds_floats(kCoarseSamples, fs.n_importance);
        const float* sorted = ray::fine_sample_ray(
            fs.rays + ray * fs.ray_stride, fs.z_c + ray * kCoarseSamples, fs.w_c + ray * kCoarseSamples,
            fs.u + ray * fs.u_row_stride, kCoarseSamples, fs.n_importance, scratch, lane_id(), mine, fs.z_f + ray * tot,
            fs.pts_f + ray * tot * 3, fs.z_samples + ray * fs.n_importance, fs.z_std + ray,
            fs.inds ? fs.inds + ray * fs.n_importance : nullptr, fs.cdf ? fs.cdf + ray * (kCoarseSamples - 1) : nullptr);
        if (wv < fs.rays_per_block)
            for (int e = lane_id(); e < tot; e += kWave) zbuf[slot * tot + e] = sorted[e];
        block_sync();                   // the depths are in place; the sampler's scratch (the park area) is free
    }
    // (the passes as straight-line code, not a loop: around a loop the register allocation of this 3 600-MFMA body falls
    //  apart -- 160 spilled registers, with one 16-byte value carried from pass to pass)
    static_for<FINE ? 3 : 1>([&](auto tile_tag) __attribute__((always_inline)) {
    constexpr int tile = decltype(tile_tag)::value;
    // (the last workgroup of an odd ray count: a pass wholly behind the last sample has no slot in the workspaces)
    if (tile < tiles && (!FINE || ((long)blockIdx.x * tiles + tile) * kSamplesPerBlock < P)) {
    if (tile > 0) block_sync();         // every wave is done with the stream buffers and the parked encoding of the last pass
    // (the fine stage's loop: what the section addresses are built from is taken as new in every pass, or the compiler
    //  hoists two dozen 64-bit section bases out of the loop into SGPRs that live, and spill, across the whole pass)
    const long Ppad = FINE ? fresh_uniform(Ppad_kernel) : Ppad_kernel;
    // (the same for the thread's LDS offsets: hoisted, every `base + k * 4096 + thread` beyond the 64 KB an LDS instruction's
    //  offset field reaches is a register of its own for the whole pass)
    const unsigned tid = FINE ? pinned_here(threadIdx.x) : threadIdx.x;
    w.tid16 = tid * 16u;
    f32x4* const park = reinterpret_cast<f32x4*>(w.lds + kStreamLds + kTableFloats * 4) + tid;
    const long wave_tile = ((long)blockIdx.x * tiles + tile) * 4 + uniform(wave_id());      // (a scalar: what derives from it -- section bases, chunk index -- is SALU work)
    // Which sample a lane works on is needed at the start (the point), before the views layer (the direction) and at the
    // end (raw, compositing): derived afresh each time from an opaque copy of the lane number -- kept alive across the
    // trunk these 64-bit indices cost six registers of a file that is full (the coarse-stage instantiation spilled).
    struct Lane { int lane, m, h; long p, pc; bool live; };
    auto lane_now = [&]() {
        Lane L;
        L.lane = (int)pinned_here((unsigned)lane_id());
        L.m = L.lane & 31;
        L.h = L.lane >> 5;
        L.p = wave_tile * kSamplesPerWave + L.m;
        L.live = L.p < P;
        L.pc = L.live ? L.p : P - 1;
        return L;
    };
    const Lane L0 = lane_now();
    const int lane = L0.lane, m = L0.m, h = (int)(pinned_here((unsigned)lane_id()) >> 5);
    const long p = L0.p, pc = L0.pc;
    const bool live = L0.live;
    (void)m;

    w.lane16 = (unsigned)lane * 16u;
    stream_prime(w.ws, wh3, w.lds, w.tid16);

    // ---- the point, its encoding (parked in LDS for the skip layer), the first operand ----
    float px, py, pz, pw = 0.f;
    auto coarse_depth_at = [&](long q) {             // the stratified depth of sample q (cheap to redo at the end)
        const float* r = cs.rays + (q >> 6) * cs.ray_stride;
        return ray::coarse_z(r[6], r[7], cs.t_vals, (int)(q & 63), kCoarseSamples, cs.lindisp, cs.t_rand != nullptr,
                             cs.t_rand ? cs.t_rand[q] : 0.f);
    };
    if constexpr (COARSE) {
        static_assert(PD == 3, "the coarse stage samples 3-D points");
        const float* r = cs.rays + (pc >> 6) * cs.ray_stride;
        const float z = coarse_depth_at(pc);
        px = r[0] + r[3] * z;
        py = r[1] + r[4] * z;
        pz = r[2] + r[5] * z;
        if (live && h == 0) {
            cs.z[p] = z;
            cs.pts[p * 3 + 0] = px; cs.pts[p * 3 + 1] = py; cs.pts[p * 3 + 2] = pz;
        }
    } else if constexpr (FINE) {
        // the point of the merged depth, as the sampler wrote it to pts_f (same expression, same rounding)
        const FwdKernelArgs mirror{pts, viewdirs, vd_stride, samples_per_ray, wpk, wh3, sc, raw, save_arg, P, cs, fs, cm};
        const auto* late = late_args(mirror);          // (read where it is used: see the compositing tail)
        const float* r = late->fs.rays + (long)((unsigned)pc / (unsigned)samples_per_ray) * late->fs.ray_stride;
        // (a lane without a sample repeats the last one, as the other stages' kernels do: pc is the clamped index)
        const float z = zbuf[(int)(pc - (long)blockIdx.x * tiles * kSamplesPerBlock)];
        px = r[0] + r[3] * z;
        py = r[1] + r[4] * z;
        pz = r[2] + r[5] * z;
    } else {
        px = pts[pc * PD + 0]; py = pts[pc * PD + 1]; pz = pts[pc * PD + 2];
        if constexpr (PD == 4) pw = pts[pc * PD + (PD - 1)];
    }
    // |encoded point| <= m_e: sines, cosines and the raw coordinates
    const float m_e = fmaxf(fmaxf(1.f, fabsf(px)), fmaxf(fmaxf(fabsf(py), fabsf(pz)), fabsf(pw)));
    u32x4 eh[NE], el[NE];
    {
        float e[ES];
        if constexpr (lab::kNoPe) {
#pragma unroll
            for (int i = 0; i < ES; ++i) e[i] = px * (float)i + py;
        } else {
            pe_slots<PD, 10, ES>(px, py, pz, pw, h, e);
        }
#pragma unroll
        for (int g = 0; g < ES / 4; ++g) park[g * kThreads] = f32x4{e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]};
        // (the saved rows from the parked copy: the wave reads back what its own lanes just wrote -- LDS is in order per wave)
        if (save) store_pe_tile<PD, 10, ES, V::kEW>(w.lds + kStreamLds + kTableFloats * 4, save + (long)kSaveEpts * Ppad, wave_tile,
                                                    uniform(wave_id()), lane);
        const float s_e = scale_for(m_e);
#pragma unroll
        for (int u = 0; u < NE; ++u) cut8(e + 8 * u, s_e, eh[u], el[u]);
    }
    block_sync();                       // the first chunk and the tables are in LDS
    ring_prime(w);

    u32x4 bh[2][16], bl[2][16];         // the two operand buffers: K slab s of a 256-wide input
    f32x16 acc[2][2];                   // two output-tile pairs
    const float* const tab_h = tables + 4 * h;
    auto section = [&](int offset, int width) {      // this wave tile's block of a tile-native section
        return uniform_global_rw(save + (long)offset * Ppad + wave_tile * (32L * width));
    };
    auto mask_block = [&](int sect) {
        unsigned* base = reinterpret_cast<unsigned*>(save + (long)V::kSavePerSample * Ppad);
        return uniform_global_rw(base + ((long)sect * (Ppad / 32) + wave_tile) * 256);
    };
    auto scale_of = [&](int layer, int what) { return sc[layer * kScaleStride + what]; };
    auto amax_of = [&](float a) { return fmaxf(a, shfl_xor(a, 32)); };
    // the largest of the wave's 32 samples -> the weight-gradient chunk this wave tile belongs to (one atomic per wave)
    // (which chunk: once per wave, as a scalar -- inside the lambda it was a 64-bit division, ~100 instructions, per layer)
    const int chunk_of_tile = (TRAIN && cm.amax) ? uniform((int)((unsigned)(wave_tile * kSamplesPerWave) / (unsigned)cm.chunk)) : 0;
    auto note_chunk_max = [&](int job, float v) __attribute__((always_inline)) {
        if constexpr (TRAIN) {
            if (cm.amax == nullptr) return;
            v = fmaxf(v, shfl_xor(v, 16)); v = fmaxf(v, shfl_xor(v, 8)); v = fmaxf(v, shfl_xor(v, 4));
            v = fmaxf(v, shfl_xor(v, 2)); v = fmaxf(v, shfl_xor(v, 1));
            if (lane_id() == 0) atomic_max_nonneg(cm.amax + (long)job * cm.n_chunks + chunk_of_tile, v);
        }
    };

    using Relu = FwdEpi<TRAIN, 0>;
    auto make_relu = [&](int l, float s_in) {         // epilogue of trunk layer l whose input was cut at s_in
        Relu e;
        e.os = inv_pow2(s_in) * scale_of(l, kSwInv);
        e.s_next = 1.f;
        e.am = 0.f;
        e.bias = tab_h + 256 * l;
        e.save = TRAIN ? section(kSaveAct + 256 * l, 256) : nullptr;
        e.lane16 = w.lane16;
        e.prime();
        return e;
    };
    auto store_mask = [&](auto& epi, int sect) {
        if constexpr (TRAIN) store_at(mask_block(sect), w.lane16, u32x4{epi.words[0], epi.words[1], epi.words[2], epi.words[3]});
    };

    // ---- layer 0: the encoded point (NE slabs) -> 256; the epilogues of pairs 0 .. 2 in the open, pair 3 under layer 1
    Relu prev = make_relu(0, scale_for(m_e));
    prev.s_next = scale_for(__builtin_fmaf(scale_of(0, kBoundA), m_e, scale_of(0, kBoundB)));
    {
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            xh = eh[s]; xl = el[s];
        };
        static_for<4>([&](auto p_tag) {
            constexpr int PP = decltype(p_tag)::value;
            tile_pair<NE * PP, NE>(w, acc[PP & 1], operand, NoFill{});
            if constexpr (PP < 3) epi_all<Relu, PP>(prev, acc[PP & 1], bh[0], bl[0]);
        });
    }

    // ---- a trunk layer: input buffer X (its tiles 6, 7 still to come from `pend`, the previous layer's last pair),
    // output buffer X ^ 1; NEF encoded-point slabs in front (the skip layer).  Leaves its own last pair pending.
    // U0: the layer's first stream unit modulo 8 (compile time).
    auto trunk_layer = [&](auto x_tag, auto nef_tag, auto u0_tag, auto& pend, auto& cur, int pend_mask_sect, int layer,
                           float am_floor, float bound_floor) __attribute__((always_inline)) {
        constexpr int X = decltype(x_tag)::value, NEF = decltype(nef_tag)::value, U0 = decltype(u0_tag)::value;
        constexpr int NK = NEF + 16;
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < NEF) { xh = eh[s]; xl = el[s]; }
            else { xh = bh[X][s - NEF]; xl = bl[X][s - NEF]; }
        };
        using Pend = std::remove_reference_t<decltype(pend)>;
        using Cur = std::remove_reference_t<decltype(cur)>;
        // pair 0 under it: the pending pair (tiles 6, 7 of the input = slabs 12 .. 15: due before slot 6 (NEF + 12))
        tile_pair<U0, NK>(w, acc[0], operand, [&](auto sg_tag) {
            epi_slot<Pend, 3, decltype(sg_tag)::value, NEF >= 4 ? 12 : 9>(pend, acc[1], bh[X], bl[X]);
        });
        if constexpr (std::is_base_of_v<MaskBits<true>, Pend>) store_mask(pend, pend_mask_sect);
        // the layer's input is complete: its measured maximum bounds this layer's output (and scales the X operand of
        // this layer's weight-gradient GEMM: job layer - 1, feature_linear: 7)
        const float am_in = amax_of(pend.am);
        note_chunk_max(layer == kLayerFeat ? 7 : layer - 1, am_in);
        const float am = fmaxf(am_in, am_floor);
        cur.s_next = scale_for(fmaxf(__builtin_fmaf(scale_of(layer, kBoundA), am, scale_of(layer, kBoundB)), bound_floor));
        tile_pair<U0 + NK, NK>(w, acc[1], operand, [&](auto sg_tag) {
            epi_slot<Cur, 0, decltype(sg_tag)::value, 12>(cur, acc[0], bh[X ^ 1], bl[X ^ 1]);
        });
        tile_pair<U0 + 2 * NK, NK>(w, acc[0], operand, [&](auto sg_tag) {
            epi_slot<Cur, 1, decltype(sg_tag)::value, 12>(cur, acc[1], bh[X ^ 1], bl[X ^ 1]);
        });
        tile_pair<U0 + 3 * NK, NK>(w, acc[1], operand, [&](auto sg_tag) {
            epi_slot<Cur, 2, decltype(sg_tag)::value, 12>(cur, acc[0], bh[X ^ 1], bl[X ^ 1]);
        });
    };

    // layers 1 .. 6: (1, 2), (3, 4), (5 = skip, 6)
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        const int l = 2 * it + 1;
        Relu cur = make_relu(l, prev.s_next);
        if (it < 2) {
            // (layer 4's output meets the encoded point in the skip layer: one scale for both)
            trunk_layer(I<0>{}, I<0>{}, I<0>{}, prev, cur, l - 1, l, 0.f, 0.f);
        } else {
            // the encoded point again, cut at the skip layer's scale
            float e[ES];
#pragma unroll
            for (int g = 0; g < ES / 4; ++g) {
                const f32x4 v = park[g * kThreads];
                e[4 * g] = v[0]; e[4 * g + 1] = v[1]; e[4 * g + 2] = v[2]; e[4 * g + 3] = v[3];
            }
#pragma unroll
            for (int u = 0; u < NE; ++u) cut8(e + 8 * u, prev.s_next, eh[u], el[u]);
            trunk_layer(I<0>{}, I<NE>{}, I<0>{}, prev, cur, l - 1, l, m_e, 0.f);
        }
        prev = cur;
        Relu cur2 = make_relu(l + 1, prev.s_next);
        trunk_layer(I<1>{}, I<0>{}, I<0>{}, prev, cur2, l, l + 1, 0.f, l + 1 == 4 ? m_e : 0.f);
        prev = cur2;
    }

    // ---- layer 7 (its epilogue also forms the density head's dot product), then the linear feature layer ----
    float vx, vy, vz;
    {
        const Lane L = lane_now();
        const long ray = (long)((unsigned)L.pc / (unsigned)samples_per_ray);      // (sample indices fit 31 bits: the workspaces of 2^31 samples would take 20 TB)
        vx = viewdirs[ray * vd_stride + 0]; vy = viewdirs[ray * vd_stride + 1]; vz = viewdirs[ray * vd_stride + 2];
    }
    const float m_ev = fmaxf(fmaxf(1.f, fabsf(vx)), fmaxf(fabsf(vy), fabsf(vz)));
    FwdEpi<TRAIN, 1> epi7;
    {
        const Relu t = make_relu(7, prev.s_next);
        epi7.os = t.os; epi7.s_next = 1.f; epi7.am = 0.f; epi7.sg = 0.f; epi7.bias = t.bias;
        epi7.alpha = tab_h + kTabAlpha; epi7.save = t.save; epi7.lane16 = t.lane16;
        epi7.prime();
    }
    trunk_layer(I<0>{}, I<0>{}, I<0>{}, prev, epi7, 6, 7, 0.f, 0.f);
    FwdEpi<TRAIN, 2> epif;
    epif.os = inv_pow2(epi7.s_next) * scale_of(kLayerFeat, kSwInv);
    epif.s_next = 1.f; epif.am = 0.f;
    epif.bias = tab_h + kTabFeat;
    epif.save = TRAIN ? section(kSaveFeat, 256) : nullptr;
    epif.lane16 = w.lane16;
    epif.prime();
    // (the feature vector meets the encoded view direction in the views layer: one scale for both)
    trunk_layer(I<1>{}, I<0>{}, I<0>{}, epi7, epif, 7, kLayerFeat, 0.f, m_ev);

    // ---- views layer on [feature (buffer 0) | encoded view direction]: 4 output tiles -> buffer 1, slabs 0 .. 7 ----
    u32x4 vh[2], vl[2];
    {
        float ev[16];
        pe_slots<3, 4, 16>(vx, vy, vz, 0.f, h, ev);
        if (save) {
            // through the park area (the encoded point's last reader was the skip layer): slot layout, then coalesced rows
            f32x4* const vpark = reinterpret_cast<f32x4*>(w.lds + kStreamLds + kTableFloats * 4) + pinned_here(threadIdx.x);
#pragma unroll
            for (int g = 0; g < 4; ++g) vpark[g * kThreads] = f32x4{ev[4 * g], ev[4 * g + 1], ev[4 * g + 2], ev[4 * g + 3]};
            store_pe_tile<3, 4, 16, 32>(w.lds + kStreamLds + kTableFloats * 4, save + (long)kSaveEviews * Ppad, wave_tile,
                                        uniform(wave_id()), (int)pinned_here((unsigned)lane_id()));
        }
        cut8(ev, epif.s_next, vh[0], vl[0]);
        cut8(ev + 8, epif.s_next, vh[1], vl[1]);
    }
    Relu epiv;
    epiv.os = inv_pow2(epif.s_next) * scale_of(kLayerViews, kSwInv);
    epiv.s_next = 1.f; epiv.am = 0.f;
    epiv.bias = tab_h + kTabViews;
    epiv.save = TRAIN ? section(kSaveHv, 128) : nullptr;
    epiv.lane16 = w.lane16;
    epiv.prime();
    if constexpr (TRAIN) { epiv.words[2] = 0u; epiv.words[3] = 0u; }
    {
        auto operand = [&](auto s_tag, u32x4& xh, u32x4& xl) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < 16) { xh = bh[0][s]; xl = bl[0][s]; }
            else { xh = vh[s - 16]; xl = vl[s - 16]; }
        };
        tile_pair<0, 18>(w, acc[0], operand, [&](auto sg_tag) {
            epi_slot<FwdEpi<TRAIN, 2>, 3, decltype(sg_tag)::value, 9>(epif, acc[1], bh[0], bl[0]);
        });
        const float am = fmaxf(amax_of(epif.am), m_ev);
        epiv.s_next = scale_for(__builtin_fmaf(scale_of(kLayerViews, kBoundA), am, scale_of(kLayerViews, kBoundB)));
        tile_pair<18, 18>(w, acc[1], operand, [&](auto sg_tag) {
            epi_slot<Relu, 0, decltype(sg_tag)::value, 12>(epiv, acc[0], bh[1], bl[1]);
        });
        epi_all<Relu, 1>(epiv, acc[1], bh[1], bl[1]);
        store_mask(epiv, 8);
    }

    // ---- rgb: one output tile over the 128 views-layer activations (stream units 36 .. 39) ----
    f32x16 accc[2];
    tile_single<36, 4>(w, accc, [&](auto s_tag, u32x4& xh, u32x4& xl) {
        constexpr int s = decltype(s_tag)::value;
        xh = bh[1][s]; xl = bl[1][s];
    }, NoFill{});
    // (+ 0, or + NaN when a parameter of the network is not finite: mlp_fwd_h3.hip's scale pass)
    const float os_rgb = __builtin_fmaf(inv_pow2(epiv.s_next), scale_of(kLayerRgb, kSwInv), scale_of(kLayerRgb, kPoison));
    const float sigma = epi7.sg + shfl_xor(epi7.sg, 32) + tables[kTabAlphaB];
    // rows 0, 1, 2 of the single tile are registers 0, 1, 2 of the h == 0 half
    const f32x4 brgb = *reinterpret_cast<const f32x4*>(tables + kTabRgb);
    const f32x4 o = {__builtin_fmaf(accc[0][0] + accc[1][0], os_rgb, brgb[0]), __builtin_fmaf(accc[0][1] + accc[1][1], os_rgb, brgb[1]),
                     __builtin_fmaf(accc[0][2] + accc[1][2], os_rgb, brgb[2]), sigma};
    const Lane LE = lane_now();
    if (LE.live && LE.h == 0) *reinterpret_cast<f32x4*>(raw + LE.p * 4) = o;
    if constexpr (FINE) {
        if (LE.h == 0) rawbuf[tile * kSamplesPerBlock + wave_id() * kSamplesPerWave + LE.m] = o;
    }
    if constexpr (COARSE) {
        // the parked-encoding area of the LDS is free (last read before the skip layer)
        float* sraw = reinterpret_cast<float*>(w.lds + kStreamLds + kTableFloats * 4);      // [128 samples][4]
        float* sz = sraw + kSamplesPerBlock * 4;                                            // [128]
        const int local = wave_id() * kSamplesPerWave + LE.m;
        if (LE.h == 0) {
            *reinterpret_cast<f32x4*>(sraw + local * 4) = o;
            sz[local] = coarse_depth_at(LE.pc);
        }
        block_sync();
        if (wave_id() < kSamplesPerBlock / kCoarseSamples) {                  // one wave per ray, lane = sample
            const int slot = wave_id();
            long ray = (long)blockIdx.x * (kSamplesPerBlock / kCoarseSamples) + slot;
            const bool ray_live = ray < cs.n_rays;
            if (!ray_live) ray = cs.n_rays - 1;
            const float norm = ray::ray_norm(cs.rays + ray * cs.ray_stride + 3);
            auto fetch = [&](int i, f32x4* rw, float* zi) {
                *rw = *reinterpret_cast<const f32x4*>(sraw + (slot * kCoarseSamples + i) * 4);
                *zi = sz[slot * kCoarseSamples + i];
            };
            ray::composite_ray(fetch, kCoarseSamples, norm, cs.noise ? cs.noise + ray * kCoarseSamples : nullptr,
                               cs.white_bkgd, ray_live, LE.lane, cs.rgb + ray * 3, cs.disp + ray, cs.acc + ray,
                               cs.depth ? cs.depth + ray : nullptr, cs.weights ? cs.weights + ray * kCoarseSamples : nullptr);
        }
    }
    }});   // (passes)
    if constexpr (FINE) {
        block_sync();                   // every pass has left its raw outputs in rawbuf
        // (what only this tail needs is read from the argument block HERE: as arguments held from the kernel's start, nine
        //  pointers' worth of SGPRs would live -- and spill -- across every pass)
        const FwdKernelArgs mirror{pts, viewdirs, vd_stride, samples_per_ray, wpk, wh3, sc, raw, save_arg, P, cs, fs, cm};
        const auto* late = late_args(mirror);
        const int rays_per_block = late->fs.rays_per_block;
        if (wave_id() < rays_per_block) {                                     // one wave per ray, 64 samples per pass
            const int slot = wave_id(), tot = kCoarseSamples + late->fs.n_importance;
            const int n_rays = late->fs.n_rays;
            long ray = (long)blockIdx.x * rays_per_block + slot;
            const bool ray_live = ray < n_rays;
            if (!ray_live) ray = n_rays - 1;
            const float norm = ray::ray_norm(late->fs.rays + ray * late->fs.ray_stride + 3);
            auto fetch = [&](int i, f32x4* rw, float* zi) {
                *rw = rawbuf[slot * tot + i];
                *zi = zbuf[slot * tot + i];
            };
            const float* noise = late->fs.noise;
            float* depth = late->fs.depth;
            float* weights = late->fs.weights;
            ray::composite_ray(fetch, tot, norm, noise ? noise + ray * tot : nullptr, late->fs.white_bkgd, ray_live, lane_id(),
                               late->fs.rgb + ray * 3, late->fs.disp + ray, late->fs.acc + ray, depth ? depth + ray : nullptr,
                               weights ? weights + ray * tot : nullptr);
        }
    }
}


template <int PD, bool TRAIN>
inline int launch_fwd_h3(const float* pts, const float* viewdirs, int vd_stride, int samples_per_ray, const float* wpacked,
                         const short* wh3, const float* scales, float* raw, float* save, long long n_samples, ChunkMaxima cm,
                         hipStream_t st) {
    constexpr unsigned lds = fwd_lds_bytes<PD>();
    SCN_LDS_OPT_IN((mlp_fwd_h3_kernel<PD, TRAIN, kPoints>), lds);
    hipLaunchKernelGGL((mlp_fwd_h3_kernel<PD, TRAIN, kPoints>), dim3(scn_ceil_div(n_samples, kSamplesPerBlock)), dim3(kThreads),
                       lds, st, pts, viewdirs, vd_stride, samples_per_ray, wpacked, wh3, scales, raw, save, (long)n_samples,
                       CoarseStage{}, FineStage{}, cm);
    return scn_launch_status();
}


template <bool TRAIN>
inline int launch_coarse_h3(const CoarseStage& cs, const float* rays, int ray_stride, const float* wpacked, const short* stream_fwd,
                            const float* scales, float* raw, float* save, ChunkMaxima cm, hipStream_t st) {
    constexpr unsigned lds = fwd_lds_bytes<3>();
    const long P = (long)cs.n_rays * kCoarseSamples;
    SCN_LDS_OPT_IN((mlp_fwd_h3_kernel<3, TRAIN, kCoarse>), lds);
    hipLaunchKernelGGL((mlp_fwd_h3_kernel<3, TRAIN, kCoarse>), dim3(scn_ceil_div(P, kSamplesPerBlock)), dim3(kThreads), lds, st,
                       (const float*)nullptr, rays + 8, ray_stride, kCoarseSamples, wpacked, stream_fwd, scales, raw, save, P, cs,
                       FineStage{}, cm);
    return scn_launch_status();
}


template <bool TRAIN>
inline int launch_fine_h3(const FineStage& fs, const float* wpacked, const short* stream_fwd, const float* scales, float* raw,
                          float* save, ChunkMaxima cm, hipStream_t st) {
    constexpr unsigned lds = fwd_lds_bytes<3, kFine>();
    const int tot = kCoarseSamples + fs.n_importance;
    const long P = (long)fs.n_rays * tot;
    SCN_LDS_OPT_IN((mlp_fwd_h3_kernel<3, TRAIN, kFine>), lds);
    hipLaunchKernelGGL((mlp_fwd_h3_kernel<3, TRAIN, kFine>), dim3(scn_ceil_div(fs.n_rays, fs.rays_per_block)), dim3(kThreads), lds,
                       st, (const float*)nullptr, fs.rays + 8, fs.ray_stride, tot, wpacked, stream_fwd, scales, raw, save, P,
                       CoarseStage{}, fs, cm);
    return scn_launch_status();
}

}  // namespace h3f
}  // namespace scn

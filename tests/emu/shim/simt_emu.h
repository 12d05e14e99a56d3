// TEST INFRASTRUCTURE -- a minimal SIMT interpreter: every GPU thread of a workgroup is a
// fiber (ucontext) on one OS thread; workgroups are distributed over a few OS threads.
// Wave-level operations (MFMA, shuffles, ballot) rendezvous the 64 fibers of a wave
// through a double-buffered exchange slot; __syncthreads rendezvous the whole workgroup.
// It executes the *same kernel sources* as the GPU build so that index arithmetic, MFMA
// fragment layouts, weight packing and barrier placement can be checked without a GPU.
// It models no timing, no bank conflicts and no memory-model weakness.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

namespace simt {

struct Idx3 { unsigned x, y, z; };

const Idx3& cur_thread_idx();
const Idx3& cur_block_idx();
const Idx3& cur_block_dim();
const Idx3& cur_grid_dim();
int cur_lane();

// Deposits `v` for this lane, waits for the rest of the wave, returns the 64 deposits.
const uint64_t* wave_exchange(uint64_t v);
void block_barrier();
char* block_lds();

void launch(Idx3 grid, Idx3 block, size_t lds_bytes, const std::function<void()>& body);

// number of OS threads used for workgroups (default: SIMT_EMU_THREADS or 8)
void set_threads(int n);

}  // namespace simt

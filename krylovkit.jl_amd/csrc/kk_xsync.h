// Level 2 of the grid reduction of the persistent MGS kernels on a row-sharded context: the sum over the RANKS, from inside
// the launch (the reference has no communication layer -- its inner products are src/orthonormal.jl:414-439
// on one address space).  Level 1 is the kernels' own single-chip reduction (grid_publish / grid_collect, panel_publish /
// panel_sweep): when it ends, wave 0 of EVERY block of a rank holds the same bits of the rank's partial.  Then
//   * block 0 stores the partial of value v as ONE tagged 16-byte granule {tag, hi, lo, tag} into slot (v, my rank) of the
//     current set of every rank's sync area (its own included) -- W stores per value, system scope (sc0 sc1), through the
//     peers' areas as hipIpcOpenMemHandle mapped them: the data IS the flag, nothing is fenced, nothing is read remotely;
//   * wave 0 of every block polls the W x nval granules of its OWN rank's area (lane = v * 8 + r, one sc0 sc1 load per pass)
//     until every tag pair equals the tag, and adds the W partials of a value with the same three-step butterfly on every rank:
//     all blocks of all ranks obtain the same bits.
// Tags count the reductions of the communicator's life (identical on all ranks: SPMD call sequence), the set is tag & 1:
// a fast rank may publish reduction t + 1 while a slow one still reads t; it cannot reach t + 2 before the slow one has
// published t + 1, i.e. has finished reading t -- across launches too, which is why the parity follows the tag and not the
// step of the launch.  ("Finished reading" holds for ALL blocks of the slow rank, not only for the publishing block 0: a cross-rank
// reduction follows the level-1 reduction of its step, which block 0 cannot complete before every block of its rank has published
// its level-1 partial, i.e. has left the poll of reduction t.)  A slot always holds the tag of the last write (tag - 2 or older): wrap-around of the 32-bit tag is
// harmless, the area is zeroed once at creation (first tags 1, 2).
// A rank that gives up (barrier timeout on its chip, test hook) stores the id of the launch into the error word of every
// peer's area: the peers stop spinning at once, nobody commits, every rank repeats the sweep on the launch-per-vector route
// (whose all-reduces then re-align the ranks).  Launch ids only grow and the word is compared MONOTONICALLY (word != 0 and
// word - launch >= 0 in 32-bit wrap-around arithmetic): with run-ahead the launch behind a lost one is already in the stream, and
// should anything of it reach a peer's word, a peer still inside the lost launch must read that as an abort of ITS launch too
// (ADVICE r5).  Only the ORIGINATOR of a give-up writes the word -- the block whose own wait ran out, a test hook -- never a block
// that merely observed the local flag or a peer's abort: a run-ahead launch that starts with the flag of its predecessor still
// raised leaves at its first reduction without touching anybody's word.  The host clears its own word before the first persistent
// launch that follows a recovery (kk_xs_launch_args; every peer's store for the lost launch is complete by then: the all-reduces
// of the repeated sweep lie in between), so a stale id cannot outlive 2^31 launches and turn into a false abort.
// A LATE rank finds the partials of the others in its area although they have given up waiting for it: a reduction therefore only
// succeeds while the abort word is clean, and the kernels ask once more (xs_aborted) before they commit -- otherwise the late rank
// would commit the sweep its peers are repeating, and the ranks' collectives would no longer pair up.
#pragma once
#include "kk_internal.h"
#include "kk_device.h"

#define KK_XS_AUX 17   // sc0 | sc1: system scope -- the granule leaves / bypasses every cache level of the issuing chip
typedef unsigned xs_v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xs_rsrc(unsigned long long base) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)base), hi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, KK_XS_BYTES, 0x00020000);
}
// tell every rank that launch `xs.launch` is lost (any lane-0 of any block; idempotent)
__device__ __forceinline__ void xs_abort(const kk_xs_dev& xs) {
    for (int r = 0; r < xs.world; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(xs.launch, xs_rsrc(xs.table[r]), KK_XS_ERR_OFFSET, 0, KK_XS_AUX);
}
// has a rank (this one included) declared launch `xs.launch` lost?
__device__ __forceinline__ bool xs_word_hits(unsigned word, unsigned launch) { return word != 0u && (int)(word - launch) >= 0; }
__device__ __forceinline__ bool xs_aborted(const kk_xs_dev& xs) {
    return xs_word_hits(__builtin_amdgcn_raw_buffer_load_b32(xs_rsrc((unsigned long long)xs.mine), KK_XS_ERR_OFFSET, 0, KK_XS_AUX), xs.launch);
}
// block 0, wave 0: the rank's partial of value v (lane v < nval) into slot (v, my rank) of every rank's area
__device__ __forceinline__ void xs_publish(const kk_xs_dev& xs, unsigned red, int nval, double local) {
    const int lane = threadIdx.x & 63;
    const unsigned tag = xs.tag0 + red;
    const unsigned set_off = (tag & 1u) * (unsigned)KK_XS_SET_BYTES;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(local);
    xs_v4u t;
    t.x = tag; t.y = (unsigned)(bits >> 32); t.z = (unsigned)bits; t.w = tag;
    for (int r = 0; r < xs.world; ++r) {   // uniform loop: one store instruction per rank, lanes v < nval active
        const __amdgpu_buffer_rsrc_t rp = xs_rsrc(xs.table[r]);
        if (lane < nval) __builtin_amdgcn_raw_buffer_store_b128(t, rp, set_off + (unsigned)(lane * KK_XS_MAX_RANKS + xs.rank) * 16u, 0, KK_XS_AUX);
    }
}
// Called by wave 0 (all 64 lanes) of every block once the rank's partials are known.  `local`: lane v < nval holds the rank's
// partial of value v (same bits in every block).  On success lane v * 8 .. v * 8 + 7 hold the total of value v; returns false
// after a timeout, a raised local flag or a peer's abort (the caller raises the local flag and leaves without committing).
// *originator (optional): the failure is THIS wave's own timeout -- only then does the caller tell the peers (xs_abort).
__device__ __forceinline__ bool xs_allreduce(const kk_xs_dev& xs, unsigned red, int nval, double local, const int* __restrict__ err, long long timeout_ticks,
                                             double& total, bool* originator = nullptr) {
    const int lane = threadIdx.x & 63;
    const unsigned tag = xs.tag0 + red;
    const unsigned set_off = (tag & 1u) * (unsigned)KK_XS_SET_BYTES;
    if (blockIdx.x == 0) xs_publish(xs, red, nval, local);
    const __amdgpu_buffer_rsrc_t rm = xs_rsrc((unsigned long long)xs.mine);
    const bool active = (lane >> 3) < nval && (lane & 7) < xs.world;
    const long long t0 = wall_clock64();
    double x = 0;
    for (;;) {
        asm volatile("" ::: "memory");   // the polls must be re-issued by every pass (see grid_collect)
        const xs_v4u g = __builtin_amdgcn_raw_buffer_load_b128(rm, set_off + (unsigned)lane * 16u, 0, KK_XS_AUX);
        const unsigned xerr = __builtin_amdgcn_raw_buffer_load_b32(rm, KK_XS_ERR_OFFSET, 0, KK_XS_AUX);
        const bool ok = !active || (g.x == tag && g.w == tag);
        const bool hit = xs_word_hits(xerr, xs.launch);
        if (__all(ok) && !hit) {
            x = active ? __longlong_as_double((long long)(((unsigned long long)g.y << 32) | g.z)) : 0.0;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        const int lerr = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool late = wall_clock64() - t0 > timeout_ticks;
        if (hit || lerr || late) {
            if (originator) *originator = !hit && !lerr;
            // post-mortem (KK_XSYNC_DEBUG=1 prints it): the first block that gave up leaves what it saw in the tail of its rank's area
            if (__builtin_amdgcn_raw_buffer_load_b32(rm, KK_XS_DBG_OFFSET, 0, KK_XS_AUX) != xs.launch) {
                __builtin_amdgcn_raw_buffer_store_b32(g.x, rm, KK_XS_DBG_OFFSET + 64u + (unsigned)lane * 4u, 0, KK_XS_AUX);
                if (lane == 0) {
                    __builtin_amdgcn_raw_buffer_store_b32(tag, rm, KK_XS_DBG_OFFSET + 4u, 0, KK_XS_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)nval, rm, KK_XS_DBG_OFFSET + 8u, 0, KK_XS_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32((hit ? 1u : 0u) | (lerr ? 2u : 0u) | (late ? 4u : 0u), rm, KK_XS_DBG_OFFSET + 12u, 0, KK_XS_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(blockIdx.x, rm, KK_XS_DBG_OFFSET + 16u, 0, KK_XS_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(red, rm, KK_XS_DBG_OFFSET + 20u, 0, KK_XS_AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(xs.launch, rm, KK_XS_DBG_OFFSET, 0, KK_XS_AUX);
                }
            }
            return false;
        }
    }
    // sum over the ranks of a value: lanes v * 8 + r, r < 8 -- butterfly over the group of 8 (every lane of the group ends with
    // the same bits: the additions of a level are commutative), the same tree on every rank
    x += dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);   // row_half_mirror: lane i <-> 7 - i of its group of 8
    total = x;
    return true;
}

// The fused per-batch step kernel and its launcher as templates over the workgroup size BS and the edge-fused update path
// (instantiated as <512, false> in step.hip, <256, false> in step256.hip, <256, true> in step256f.hip: translation units
// that build in parallel): the readout of (src,dst) and
// (src,neg) on the pre-batch state and the update of the batch's targets in ONE launch
// (train_link_prediction.py:325-373 order; models/TPNet.py:67-99, :112-128).
#pragma once
#include "readout.hpp"
#include "update.hpp"

namespace tpnet {

// row sharding: does this rank compute node `id`'s update / the pairs whose src it is?  own_mod > 1: cyclic ownership over the
// global ids (id % own_mod == own_rem); own_mod == 0: compact local tables, owned rows first (id < own_rem; the rows behind
// them are the batch's halo: copies of other ranks' rows); own_mod == 1: everything
__device__ __forceinline__ bool owns(const StreamArgs& a, int64_t id) {
    if (a.own_mod > 1) return (int32_t)((uint64_t)id % (uint32_t)a.own_mod) == a.own_rem;
    if (a.own_mod == 0) return id < (int64_t)a.own_rem;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// fused per-batch step: readout (src,dst) and (src,neg) on the pre-batch state + update, ONE launch.
// Blocks [0, HEAVY_BLOCKS) take the heavy update items (one workgroup per target and layer); the others walk a work index
// space: [0, RP) readout pairs (RP rounded up to whole waves so that a wave has one role), then the light items.
// ---------------------------------------------------------------------------------------------------------------
template <int LPP, int VPL, int W, int L, bool FULL, bool NT, int BS, bool FUSE>
// The arguments every wave needs FIRST (the id arrays, this batch's item lists, e0/ne, the role flags incl. which outputs
// exist) lead the signature as plain scalars: the first 16 SGPRs of kernel arguments are preloaded by the command
// processor at wave launch (-mllvm -amdgpu-kernarg-preload-count=16), so the role of a wave is decided and its id / item
// loads are issued without waiting for a kernarg fetch.
__global__ __launch_bounds__(BS, step_min_waves(LPP, VPL, W, BS)) void k_step(const int64_t* __restrict__ a_src, const int64_t* __restrict__ a_dst,
                                                const int64_t* __restrict__ a_neg, const Item* __restrict__ items,
                                                const Item* __restrict__ heavy, uint32_t e0_, int32_t ne_,
                                                uint32_t flags, int HEAVY_BLOCKS, uint32_t bid, double lambda,
                                                tpnet_state S, StreamArgs a, Plan p, int64_t b) {
    const int64_t e0 = e0_;     // first edge of the batch inside the chunk (< 2^30: api.hip caps a chunk there); 14 SGPRs so far
    constexpr int GPB = BS / LPP;
    constexpr int GPW = 64 / LPP;
    constexpr int NG = GramCfg<LPP, L>::NG;
    constexpr int STG = GramCfg<LPP, L>::template stage_floats<BS>();
    __shared__ float part[(VPL * W * BS > STG) ? VPL * W * BS : STG];   // heavy items' partial sums / readout staging
    // (round 4, measured and not kept: the multi-pass launches' NEXT pair fetched ahead -- its ids two passes early into registers,
    // its two meta records one pass early straight into LDS with global_load_lds_dwordx4, so that a pass starts with its row loads:
    // bit-identical results, but the kernel sits at the 168 registers three waves per SIMD allow, the pipeline's state spilled
    // 17 of them, and a batch of 10 000 edges took 52.4 us instead of 46.5 on the C4 table, 40.4 instead of 32.6 at C3;
    // round 5, the lightest form of it -- only the NEXT pair's two ids and its edge-fused flags fetched while the current pair is computed:
    // scratch 12 -> 40 bytes per lane, 46.9 us against 44.8 on the C4 table, 35.5 against 32.5 at C3: not kept either.  What IS kept from
    // round 5: the layer-0 rows of a pair, whose addresses need no record, are requested with the records (readout.hpp: EARLY0) -- C4
    // 44.4 -> 43.3 us, C3 / C5 unchanged)
    unsigned long long* dbg = p.dbg;
    (void)dbg;
    STAMP(0);
    // e0/ne come with the launch and the item records are fetched speculatively (their slots always exist), so neither
    // the id loads nor the item loads wait for the batch descriptor (clocks, item counts).  The descriptor is read
    // AFTER those vector loads have been issued: it is a scalar load whose wait (lgkmcnt) would otherwise sit in front
    // of them and put one more memory round trip on every wave's critical path.
    const BatchDesc* __restrict__ Dp = p.desc + b;
    // Reserved workgroups without a hub unit (a batch with few or no hubs: HEAVY_BLOCKS is a launch parameter, the number of
    // hubs a device-side fact) do not idle: reserved workgroup number units + k takes overflow chunk k, and the regular
    // workgroups skip the chunks taken this way in their later rounds.  Both sides decide from the batch descriptor alone
    // (no atomics, no hand-off); a regular workgroup's FIRST unit is untouched, so the one-pass batches the grid is sized for
    // run exactly as before.  Only the small-batch kernels (no STEP_ITEMS_FIRST remap) use it.
    int surplus = -1;
    if ((int)blockIdx.x < HEAVY_BLOCKS) {
        if (flags & ROLE_UPDATE) {
            // heavy work unit = (item, layer, column part): the layers of the update are independent sums and so are the
            // columns of a row.  In the 256-thread kernels (large batches: one item of a C3 / C5 batch has ~1 000
            // contributions) a row group of 32 / 64 lanes is cut into parts of 16 lanes, so that 16 slices instead of 8 / 4
            // share the hub's list inside a workgroup and 2 / 4 times as many workgroups share the hub.
            constexpr int CP = (BS == BLOCK_SMALL && FULL && W == 4 && LPP > 16) ? LPP / 16 : 1;
            constexpr int LPH = LPP / CP;
            const uint32_t cap = 2u * (uint32_t)ne_ * (uint32_t)(L * CP);
            for (uint32_t h = blockIdx.x; h < cap; h += HEAVY_BLOCKS) {
                const Item I = heavy[h / (L * CP)];
                const uint32_t n_heavy = Dp->n_heavy;
                const double t_last = Dp->t_last;
                if (h / (L * CP) >= n_heavy) {
                    // a reserved workgroup that has NO hub unit at all takes one chunk of the OVERFLOW of the work index
                    // space (what the other workgroups would walk in a second grid-stride round): see `surplus` below
                    if (h == blockIdx.x) surplus = (int)(blockIdx.x - n_heavy * (uint32_t)(L * CP));
                    break;
                }
                if (!owns(a, I.target)) continue;
                update_item_block<LPH, VPL, W, L, FULL, NT, BS>(S, p.s_partner, p.s_coef, I.target, I.j0, I.cnt,
                                                                (int)((h / CP) % L), bid, t_last, lambda, part,
                                                                (int)(h % CP) * LPH * VPL);
                STAMP(7);
            }
        }
        if (surplus < 0 || FUSE || !(flags & ROLE_UPDATE)) return;
    }
    const int gl = threadIdx.x % LPP;
    const int g = threadIdx.x / LPP;
    const bool packed = (flags & TPNET_FLAG_PACKED) != 0;
    const bool do_scale = !(flags & TPNET_FLAG_NOT_SCALE) && !packed;
    const int ostride = packed ? GramCfg<LPP, L>::NT : NG;
    const int64_t ne = ne_;
    const int64_t npos = ((flags & ROLE_READOUT) && (flags & STEP_HAS_POS)) ? ne : 0;
    const int64_t nneg = ((flags & ROLE_READOUT) && (flags & STEP_HAS_NEG)) ? ne : 0;
    // (a shared-src unit per edge -- gram_shared -- was measured here: it halves the readout waves but doubles each
    // wave's VALU chain: C1 +30 %, C2 +5 % slower, C3/C5 +1 % faster; the pair stays the unit)
    const int64_t npairs = npos + nneg;
    const int64_t RP = (npairs + GPW - 1) / GPW * GPW;
    const int64_t cap_items = (flags & ROLE_UPDATE) ? 2 * ne : 0;   // upper bound of the light items (slots exist)
    // narrow rows: an item takes 16 lanes = ISL group slots of the work index space (update_item_narrow)
    constexpr int ISL = (LPP < 16 && W == 4) ? 16 / LPP : 1;
    const int64_t total = RP + cap_items * ISL;
    const int64_t nblk = (int64_t)gridDim.x - HEAVY_BLOCKS;

    // STEP_ITEMS_FIRST (large batches): the light items lead the work index space instead of trailing it.  An item is a
    // longer chain of dependent round trips than a pair (up to heavy_threshold contributions, a few rows in flight), so
    // it should start first and leave the short pairs to fill the end of the launch.  Needs the batch's item count up
    // front: one scalar load that a small, latency-bound batch does not want to wait for.
    // (compiled into the fused-plan variant only -- the same batches take both -- so that the other variants keep their
    // prologue of ONE scalar-load round trip: with this block present the compiler splits the kernel-argument loads over
    // two waits, +0.3 us on every wave of a small batch)
    int64_t lead = 0, total_w = total;
    if constexpr (FUSE) {
        if (flags & STEP_ITEMS_FIRST) {
            lead = (((int64_t)Dp->n_light * ISL + GPW - 1) / GPW) * GPW;
            total_w = lead + RP;
        }
    }

    // overflow chunk c (c >= 0) = work indices [(nblk + c) * GPB, ...): taken by reserved workgroup (hub units + c) if that
    // one exists.  first_free = the first overflow chunk NOT taken by a reserved workgroup.
    int64_t base0 = ((int64_t)blockIdx.x - HEAVY_BLOCKS) * GPB;
    int64_t round_stride = nblk * GPB;
    int64_t first_free = 0;
    if constexpr (!FUSE) {
        if (surplus >= 0) {
            base0 = (nblk + surplus) * (int64_t)GPB;       // exactly one chunk
            round_stride = total_w;                        // (leaves the loop after it)
        } else if (flags & ROLE_UPDATE) {
            constexpr int CPh = (BS == BLOCK_SMALL && FULL && W == 4 && LPP > 16) ? LPP / 16 : 1;
            // (only a workgroup that HAS a later round needs the hub count: read behind its first unit's loads)
            first_free = -1;                               // decided lazily below
            (void)CPh;
        }
    }
    for (int64_t base = base0; base < total_w; base += round_stride) {
        if constexpr (!FUSE) {
            if (surplus < 0 && (flags & ROLE_UPDATE) && base != base0) {
                // later rounds of a regular workgroup: chunks [nblk, nblk + idle reserved workgroups) are gone
                constexpr int CPh = (BS == BLOCK_SMALL && FULL && W == 4 && LPP > 16) ? LPP / 16 : 1;
                if (first_free < 0) {
                    const int64_t units = (int64_t)Dp->n_heavy * (L * CPh);
                    first_free = units < HEAVY_BLOCKS ? (int64_t)HEAVY_BLOCKS - units : 0;
                }
                const int64_t chunk = base / GPB - nblk;   // overflow chunk number of this round (>= 0)
                if (chunk < first_free) continue;
            }
        }
        int64_t w = base + g;
        int64_t wave0 = base + (g / GPW) * GPW;        // first work index of this wave: decides the wave's role
        if constexpr (FUSE) {
            if (flags & STEP_ITEMS_FIRST) {            // positions [0, lead) are the items, [lead, lead + RP) the pairs
                const int64_t sh = (wave0 < lead) ? RP : -lead;
                w += sh;
                wave0 += sh;
            }
        }
        if (wave0 < RP) {
            const bool valid = w < npairs;
            int64_t e = 0, u = 0, v = 0;
            float* out = nullptr;
            bool a_dst_pair = false;
            if (valid) {
                // (src,dst) and (src,neg) of one edge sit in ADJACENT lane groups (work index 2e, 2e+1): the two groups
                // load the src node's meta record and rows with the same instructions and the same addresses, which the
                // load unit coalesces -- one fetch from the memory side instead of two in different workgroups/XCDs
#ifndef TPNET_NO_INTERLEAVE
                const bool both = npos != 0 && nneg != 0;
#else
                const bool both = false;
#endif
                const int64_t idx = both ? (w >> 1) : (w < npos ? w : w - npos);
                const bool isneg = both ? (w & 1) != 0 : (w >= npos);
                e = e0 + idx;
                a_dst_pair = !isneg;
                out = (isneg ? a.out_neg : a.out_pos) + e * ostride;
                v = isneg ? a_neg[e] : a_dst[e];
                u = a_src[e];
            }
            // edge-fused update: the (src,dst) pair of an edge also writes the bundles of those endpoints whose only
            // contribution in this batch is this edge (Plan::fuse_*)
            constexpr bool LR = GramCfg<LPP, L>::template lds_reduce<BS>();
            constexpr bool CAN_FUSE = FUSE;                // a kernel variant of its own: the unfused ones stay lean
            uint32_t fbits = 0;
            float fw = 0.0f;
            if (CAN_FUSE && (flags & STEP_FUSE) && valid && a_dst_pair) {
                fbits = (uint32_t)p.fuse_src[e] | ((uint32_t)p.fuse_dst[e] << 1);
                fw = p.fuse_w[e];
            }
            const double now = Dp->now;
            const double t_last_p = Dp->t_last;
            // row sharding: a pair is read out by the owner of its src node (the other ranks leave the output row alone)
            const bool mine = valid && owns(a, u);
            if (!__any(mine)) continue;
            if (packed)
                gram_pair<LPP, VPL, W, L, FULL, true, CAN_FUSE, LR>(S, u, v, mine, bid, now, lambda, do_scale, out, gl, p.dbg, part, fbits,
                                                      fw, t_last_p);
            else
                gram_pair<LPP, VPL, W, L, FULL, false, CAN_FUSE, LR>(S, u, v, mine, bid, now, lambda, do_scale, out, gl, p.dbg, part, fbits,
                                                       fw, t_last_p);
            STAMP(5);
        } else {
            // (a launch WITHOUT the update role -- the exact mode's readout launch, api.hip -- has no items: the lane groups behind
            // the last pair of its last workgroup land here.  Until round 5 they walked the first items of the batch's list: an
            // update before the dense decay, formed again by the update launch from the copy the decay had then not scaled --
            // results inside the exact mode's tolerance, but not the same bits run to run)
            if (!(flags & ROLE_UPDATE)) break;
            const int64_t it = (w - RP) / ISL;
            Item I = items[it < cap_items ? it : 0];
            const int64_t n_light = (int64_t)Dp->n_light;
            const double t_last = Dp->t_last;
            if ((wave0 - RP) / ISL >= n_light) break;   // wave-uniform: no item of this wave exists (and none later)
            const bool valid = it < n_light && owns(a, I.target);   // targets belong to their owner
            if constexpr (ISL > 1)
                update_item_narrow<LPP, L>(S, p.s_partner, p.s_coef, I, valid, bid, t_last, lambda, (int)(threadIdx.x % 16));
            else
                update_item<LPP, VPL, W, L, FULL, NT>(S, p.s_partner, p.s_coef, I, valid, bid, t_last, lambda, gl);
            STAMP(6);
        }
    }
}


// resident workgroups of a kernel on this device: occupancy (per CU, from the runtime) x CU count
template <typename K>
static int resident_blocks(K kernel, int block_threads) {
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return cus * per_cu;
}


template <int BS, bool FUSE>
int launch_step_bs(const tpnet_state& st, const StreamArgs& a, const Plan& p, int64_t b, int64_t batch, int32_t ne,
                   double lambda, uint32_t launch_id, uint32_t flags, hipStream_t s) {
    // narrow rows (d <= 32): 4 / 8 lanes per row with 16-lane items, whose sums are not in index order -- a
    // TPNET_FLAG_SEQUENTIAL launch keeps the 16-lane geometry
    if ((reinterpret_cast<uintptr_t>(a.out_pos) | reinterpret_cast<uintptr_t>(a.out_neg)) & 15) return TPNET_ERR_BAD_ARG;
    Geom geom = pick_geom(st.d);
    if ((flags & TPNET_FLAG_SEQUENTIAL) && geom.w == 4 && geom.lpp < 16) geom = Geom{16, 1, 4};
    TPNET_DISPATCH_G(geom, ({
        constexpr int GPB = BS / LPP;
        constexpr int ISL = (LPP < 16 && W == 4) ? 16 / LPP : 1;
        static const int resident = resident_blocks(k_step<LPP, VPL, W, L, FULL, false, BS, FUSE>, BS);
        static const int hb_env = TPNET_DEV_INT(HEAVY_BLOCKS, 0);
        const int HEAVY_BLOCKS = hb_env > 0 ? hb_env : (ne <= 1024 ? HEAVY_BLOCKS_SMALL : HEAVY_BLOCKS_LARGE);
        // One pass when it fits: 2*ne readout pairs + up to 2*ne distinct targets.  A small batch is bound by its
        // chain of dependent memory round trips, so every workgroup should be resident at once (a workgroup that
        // starts after another one has finished doubles the chain) -- the item slots give way first (a batch
        // rarely has more than ne distinct light targets; the grid-stride loop covers the rest).
        const int pair_blocks = grid_for(2 * (int64_t)ne, GPB, 1 << 20);
        int item_blocks = grid_for(2 * (int64_t)ne * ISL, GPB, 1 << 20);
        const int room = resident - HEAVY_BLOCKS - pair_blocks;
        const int item_min = grid_for((((int64_t)ne * 3) / 4 + 1) * ISL, GPB, 1 << 20);
        if (item_blocks > room) item_blocks = room > item_min ? room : item_min;
        int grid = HEAVY_BLOCKS + pair_blocks + item_blocks;
        static const int cap_env = TPNET_DEV_INT(GRID_CAP, 0);
        const int GRID_CAP = cap_env > 0 ? cap_env : 256 * 8;
        if (grid > HEAVY_BLOCKS + GRID_CAP) grid = HEAVY_BLOCKS + GRID_CAP;
        // streamed state stores for mid-size batches on the two geometries that serve them (see stv)
        static const char* nt_env = TPNET_DEV_STR(NT_STATE);          // developer override: "0" / "1"
        constexpr bool NT_GEOM = BS == BLOCK && W == 4 && VPL == 1 && (LPP == 16 || LPP == 32);
        const bool nt_state = NT_GEOM && (nt_env ? nt_env[0] == '1' : (ne >= 512 && ne <= 2048));
        static const char* if_env = TPNET_DEV_STR(ITEMS_FIRST);        // developer override: "0" / "1"
        const bool items_first = FUSE && (flags & ROLE_UPDATE) && (flags & ROLE_READOUT) && (if_env ? if_env[0] == '1' : ne > 1024);
        const uint32_t kflags = flags | (items_first ? STEP_ITEMS_FIRST : 0u) | (a.out_pos ? STEP_HAS_POS : 0u) |
                                (a.out_neg ? STEP_HAS_NEG : 0u);
        const Item* light0 = p.light + 2 * (b * batch);
        const Item* heavy0 = p.heavy + 2 * (b * batch);
        if (nt_state)
            hipLaunchKernelGGL((k_step<LPP, VPL, W, L, FULL, NT_GEOM, BS, FUSE>), dim3(grid), dim3(BS), 0, s, a.src, a.dst,
                               a.neg, light0, heavy0, (uint32_t)(b * batch), ne, kflags, HEAVY_BLOCKS, launch_id, lambda, st, a, p, b);
        else
            hipLaunchKernelGGL((k_step<LPP, VPL, W, L, FULL, false, BS, FUSE>), dim3(grid), dim3(BS), 0, s, a.src, a.dst,
                               a.neg, light0, heavy0, (uint32_t)(b * batch), ne, kflags, HEAVY_BLOCKS, launch_id, lambda, st, a, p, b);
    }));
    TPNET_HIP_TRY(hipGetLastError());
    return TPNET_OK;
}


}  // namespace tpnet

// gn2v_train_world: the block fit of several ranks -- one process (or thread) per GPU -- driven
// from C through a communicator the host fills (include/gn2v.h gn2v_comm: RCCL, MPI,
// torch.distributed behind callbacks, an in-process loop-back).  The schedule is the one
// embiggen_amd/distributed.py's BlockPartitionedTrainer.train_round runs for world > 1 (and its
// docstring states): the central table striped over the ranks, the contextual table in
// parts = P x world travelling parts, per round the walks all-gathered, every rank extracting the
// pairs whose centre it owns a group of parts at a time, one gn2v_block_step per part while the
// part finished last episode leaves for rank - 1 and the part needed P - 1 episodes from now
// arrives from rank + 1.  Replaces, for a non-Python binding, the multi-GPU form of
// `self._model.fit_transform(graph)` (embedders/ensmallen_embedders/node2vec.py:99).
//
// Included at the end of gn2v_block_api.hip (it uses that unit's Buffers and planning helpers).
#pragma once

namespace {

// rows of `src` (f32[rows][ld]) -> rows first, first + stride, ... of `dst` (f32[...][ld])
int scatter_rows(float *dst, const float *src, uint64_t rows, uint32_t ld, uint64_t first,
                 uint64_t stride, hipStream_t s) {
    if (rows == 0) return 0;
    HIP_TRY(hipMemcpy2DAsync(dst + first * ld, (size_t)stride * ld * 4, src, (size_t)ld * 4,
                             (size_t)ld * 4, rows, hipMemcpyDeviceToDevice, s));
    return 0;
}

#define COMM_TRY(expr, what)                                                                  \
    do {                                                                                      \
        if ((expr) != 0)                                                                      \
            return fail(std::string("gn2v_train_world: the communicator's ") + (what) +       \
                        " failed");                                                           \
    } while (0)

}  // namespace

extern "C" int gn2v_train_world(gn2v_graph *g, const gn2v_walk_params *wp,
                                const gn2v_train_params *tp, uint64_t seed,
                                uint64_t max_walks_per_epoch, uint64_t round_walks,
                                const gn2v_comm *comm, float *d_central, float *d_contextual,
                                gn2v_stats *stats, void *stream) {
    if (!g || !wp || !tp || !comm) return fail("NULL handle / params / communicator");
    if (tp->model != GN2V_MODEL_SKIPGRAM) return fail("the block path trains SkipGram only");
    if (!d_central || !d_contextual) return fail("NULL table pointer");
    if (comm->world < 1 || comm->rank >= comm->world) return fail("need rank < world");
    if (!comm->all_gather || !comm->sendrecv_start || !comm->sendrecv_wait || !comm->broadcast)
        return fail("the communicator must provide all_gather, sendrecv_start, sendrecv_wait "
                    "and broadcast");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t rank = comm->rank, world = comm->world;
    const uint64_t n = g->view.n_nodes;
    const uint32_t L = wp->walk_length, w = tp->window, ld = tp->ld;

    if (prepare_walk_sampler(g, wp, s)) return 1;

    // ---- plan (the rule of gn2v_train_blocks and of the Python trainer)
    gn2v_block_plan plan{};
    plan.world = world;
    plan.rank = rank;
    if (gn2v_block_auto_plan_graph(g, world, ld, tp->k, &plan.parts, &plan.slices, s)) return 1;
    if (world > 1 && (plan.parts % world || plan.parts < 2 * world))
        return fail("context parts must be a multiple of the ranks, at least two per rank");
    plan.walk_length = L;
    plan.window = w;
    plan.min_dist = tp->min_dist ? tp->min_dist : 1;
    plan.record = 0;
    for (uint32_t r = 32; r >= 8 && !plan.record; r >>= 1)
        if (block_lds_words_per_wave(ld, r, tp->k) * 4 * (gn2v::kTrainBlock / 64) <= 64 * 1024)
            plan.record = r;
    if (!plan.record) return fail("number_of_negative_samples beyond the block path's LDS plans");
    plan.flags = tp->flags & GN2V_TRAIN_DOWNSAMPLE;
    const bool resident_plan = plan.slices > gn2v_host::kCursorSlices;
    const bool permute = resident_plan && env_size("GN2V_BLOCK_PERMUTE", 1) != 0;
    plan.hot_rows = resident_plan ? 0 : (uint32_t)env_size("GN2V_HOT_ROWS", GN2V_BLOCK_HOT_DEFAULT);
    plan.hot_flush = (uint32_t)env_size("GN2V_HOT_FLUSH", 0);
    if (gn2v_block_plan_check(g, &plan)) return 1;
    const uint32_t parts = plan.parts, cells = parts * plan.slices, per_rank = parts / world;
    const bool scale_free = tp->flags & GN2V_TRAIN_SCALE_FREE;
    for (uint32_t p = 0; p < parts; ++p)
        if (gn2v::stripe_count(n, p, parts) == 0)
            return fail("a context part owns no node: graph too small to split this far");

    Buffers buf(g);
    // ---- tables of the negatives (fixed cells: once; under a placement: every round)
    uint64_t *alias = nullptr, *cell_rows = nullptr;
    uint32_t *hub_bits = nullptr, *hot_list = nullptr, *place = nullptr, *inv = nullptr;
    uint8_t *hot_slot = nullptr;
    void *alias_tmp = nullptr, *place_tmp = nullptr;
    uint64_t alias_tb = 0, place_tb = 0;
    if (scale_free || plan.hot_rows) {
        gn2v_block_alias_temp_bytes(n, &alias_tb);
        if (buf.alloc(&alias, n * 8) || buf.alloc(&cell_rows, (cells + 1) * 8) ||
            buf.alloc(&alias_tmp, alias_tb))
            return 1;
        if (plan.hot_rows &&
            (buf.alloc(&hub_bits, ((n + 31) / 32) * 4) ||
             buf.alloc(&hot_list, (size_t)cells * GN2V_BLOCK_HOT_MAX * 4) || buf.alloc(&hot_slot, n)))
            return 1;
        if (!permute && gn2v_block_alias(g, &plan, alias, cell_rows, hub_bits, hot_list, hot_slot,
                                         nullptr, alias_tmp, alias_tb, s))
            return 1;
    }
    if (permute) {
        if (gn2v_block_placement_temp_bytes(n, &place_tb)) return 1;
        if (buf.alloc(&place, n * 4) || buf.alloc(&inv, n * 4) || buf.alloc(&place_tmp, place_tb))
            return 1;
    }

    // ---- this rank's shards: its central partition (never moves) and P + 1 part buffers
    const uint64_t my_rows = gn2v::stripe_count(n, rank, world);
    const uint64_t max_part_rows = gn2v::stripe_count(n, 0, parts);
    float *central = nullptr;
    if (buf.alloc(&central, std::max<uint64_t>(1, my_rows) * ld * 4)) return 1;
    if (gn2v_init_table_rows(central, my_rows, tp->d, ld, seed, 0, tp->init_scale, rank, world, s))
        return 1;
    std::vector<float *> held(parts, nullptr);  // part id -> the buffer that holds it here
    for (uint32_t i = 0; i < per_rank; ++i) {
        const uint32_t p = per_rank * rank + i;
        if (buf.alloc(&held[p], max_part_rows * ld * 4)) return 1;
        if (gn2v_init_table_rows(held[p], gn2v::stripe_count(n, p, parts), tp->d, ld, seed, 1,
                                 tp->init_scale, p, parts, s))
            return 1;
    }
    float *spare = nullptr;
    if (world > 1 && buf.alloc(&spare, max_part_rows * ld * 4)) return 1;

    // ---- round size and groups: every rank the same (the minimum of what each would take)
    uint64_t walks_per_epoch = g->view.n_sources * (uint64_t)wp->iterations;
    if (max_walks_per_epoch && max_walks_per_epoch < walks_per_epoch)
        walks_per_epoch = max_walks_per_epoch;
    uint64_t *agree = nullptr, *agreed = nullptr;
    if (buf.alloc(&agree, 16) || buf.alloc(&agreed, 16 * (size_t)world)) return 1;
    uint32_t group_parts = 0;
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        {
            std::lock_guard<std::mutex> lock(g->kept_mu);
            free_b += g->kept_bytes;
        }
        uint64_t mine[2] = {round_walks, 0};  // the plan's cap: the caller's round, or ...
        if (round_walks == 0 && permute) {  // ... 16-64 rounds an epoch (gn2v_train_blocks: the same rule)
            const uint64_t rounds = gn2v_host::rounds_per_epoch(tp->epochs);
            const uint64_t epoch = g->view.n_sources * (uint64_t)wp->iterations;
            // (several ranks: a training launch is ONE part of one rank -- round x 1 250 / parts
            // pairs -- so a rank's round keeps at least 2^19 walks: a rank of 8 on the bench graph
            // ran at 1.73e9 pairs/s with rounds of 194 k walks, at 1.88e9 with 874 k)
            const uint64_t shortest = std::max<uint64_t>(
                1, env_size("GN2V_ROUND_MIN_WALKS", world > 1 ? 1ull << 19 : 1ull << 14));
            mine[0] = std::max<uint64_t>(shortest, (epoch + rounds * world - 1) / (rounds * world));
        }
        if (gn2v_block_round_plan(free_b, n, L, w, world, parts, plan.slices, 0, &mine[0],
                                  &group_parts))
            return 1;
        mine[1] = group_parts;
        HIP_TRY(hipMemcpyAsync(agree, mine, 16, hipMemcpyHostToDevice, s));
        COMM_TRY(comm->all_gather(comm->ctx, agree, agreed, 16, stream), "all_gather");
        std::vector<uint64_t> all(2 * (size_t)world);
        HIP_TRY(hipMemcpyAsync(all.data(), agreed, 16 * (size_t)world, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t r = 0; r < world; ++r) {
            mine[0] = std::min(mine[0], all[2 * r]);
            mine[1] = std::min(mine[1], all[2 * r + 1]);
        }
        if (round_walks == 0) round_walks = mine[0];
        group_parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(mine[1], parts));
    }
    round_walks = std::max<uint64_t>(1, std::min(round_walks, (walks_per_epoch + world - 1) / world));
    const uint64_t stride = (uint64_t)world * round_walks;  // walks of a round, all ranks
    const uint64_t pairs_per_walk = 2ull * w * L;

    uint32_t *walks = nullptr, *walks_all = nullptr, *placed = nullptr;
    uint64_t *pairs = nullptr, *work = nullptr, *cell_offsets = nullptr;
    void *tmp = nullptr;
    uint64_t cap = round_walks * world * pairs_per_walk / world / parts * group_parts;
    cap += cap / 8 + 1024;
    uint64_t tb = 0;
    gn2v_block_extract_temp_bytes(cap, &tb);
    if (buf.alloc(&work, GN2V_BLOCK_WORK_WORDS * 8) || buf.alloc(&cell_offsets, (cells + 1) * 8) ||
        buf.alloc(&walks, round_walks * L * 4) ||
        (world > 1 && buf.alloc(&walks_all, stride * L * 4)) ||
        (permute && buf.alloc(&placed, stride * L * 4)) || buf.alloc(&pairs, cap * 8) ||
        buf.alloc(&tmp, tb))
        return 1;
    if (world == 1) walks_all = walks;

    // ---- rounds
    auto part_of_episode = [&](uint64_t e) { return (uint32_t)((per_rank * rank + e) % parts); };
    uint64_t episode = 0, round_id = 0, trained = 0;
    float lr = tp->lr;
    for (uint32_t e = 0; e < tp->epochs; ++e) {
        for (uint64_t first = 0; first < walks_per_epoch; first += stride, ++round_id) {
            // this rank's walks of the round (ranks with fewer left pad with ended walks)
            const uint64_t mine = first + (uint64_t)rank * round_walks;
            const uint64_t nw = mine < walks_per_epoch
                                    ? std::min(round_walks, walks_per_epoch - mine) : 0;
            if (nw < round_walks)
                HIP_TRY(hipMemsetAsync(walks, 0xFF, round_walks * L * 4, s));
            if (nw && gn2v_walks(g, wp, seed, e, mine, nw, walks, s)) return 1;
            if (world > 1)
                COMM_TRY(comm->all_gather(comm->ctx, walks, walks_all, round_walks * L * 4, stream),
                         "all_gather");
            if (permute) {  // this round's cells: a row never leaves its part (classes = parts,
                            // with one rank too: the parts are separate buffers here)
                if (gn2v_block_placement(g, parts, seed, round_id, place, inv, place_tmp, place_tb,
                                         s))
                    return 1;
                if (alias && gn2v_block_alias(g, &plan, alias, cell_rows, nullptr, nullptr, nullptr,
                                              inv, alias_tmp, alias_tb, s))
                    return 1;
                if (gn2v_block_place_walks(place, walks_all, stride * L, placed, s)) return 1;
            }
            // groups of parts in this rank's episode order (a round starts at an episode that is
            // a multiple of `parts`)
            for (uint32_t e0 = 0; e0 < parts; e0 += group_parts) {
                const uint32_t p0 = part_of_episode(e0), pn = std::min(group_parts, parts - e0);
                if (gn2v_block_count(g, &plan, walks_all, placed, stride, seed, e, first, p0, pn,
                                     work, cell_offsets, s))
                    return 1;
                uint64_t n_pairs = 0;
                HIP_TRY(hipMemcpyAsync(&n_pairs, cell_offsets + cells, 8, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                uint64_t need = 0;
                gn2v_block_extract_temp_bytes(n_pairs, &need);
                if (n_pairs > cap || need > tb) {  // a group heavier than the head room: grow
                    buf.free_last();
                    buf.free_last();
                    cap = n_pairs + n_pairs / 16;
                    gn2v_block_extract_temp_bytes(cap, &tb);
                    if (buf.alloc(&pairs, cap * 8) || buf.alloc(&tmp, tb)) return 1;
                }
                if (n_pairs) {
                    if (gn2v_block_extract(g, &plan, walks_all, placed, stride, seed, e, first, p0,
                                           pn, work, hub_bits, n_pairs, pairs, tmp, tb, s))
                        return 1;
                    if (gn2v_block_cell_offsets(g, &plan, pn, pairs, n_pairs, cell_offsets, s))
                        return 1;
                }
                for (uint32_t i = 0; i < pn; ++i, ++episode) {
                    const uint32_t part = part_of_episode(episode);
                    void *hop = nullptr;
                    uint32_t done = 0, nxt = 0;
                    const bool hops = world > 1 && episode >= 1;
                    if (hops) {
                        // the part finished last episode leaves for rank - 1, the part needed
                        // per_rank - 1 episodes from now arrives from rank + 1, both while this
                        // episode trains
                        done = part_of_episode(episode - 1);
                        nxt = part_of_episode(episode + per_rank - 1);
                        COMM_TRY(comm->sendrecv_start(
                                     comm->ctx, held[done], gn2v::stripe_count(n, done, parts) * ld * 4,
                                     (rank + world - 1) % world, spare,
                                     gn2v::stripe_count(n, nxt, parts) * ld * 4, (rank + 1) % world,
                                     stream, &hop),
                                 "sendrecv_start");
                    }
                    if (n_pairs) {
                        gn2v_block_io io{};
                        io.d_pairs = pairs;
                        io.d_cell_offsets = cell_offsets;
                        io.d_alias = scale_free ? alias : nullptr;
                        io.d_cell_rows = cell_rows;
                        io.d_hot_list = hot_list;
                        io.d_hot_slot = hot_slot;
                        io.d_central = central;
                        io.d_context = held[part];
                        io.block_id = round_id * world + rank;
                        io.part = part;
                        io.d_inv = permute ? inv : nullptr;  // rows at d_context + (x / parts) ld
                        if (gn2v_block_step(g, tp, &plan, &io, seed, e, lr, s)) return 1;
                    }
                    if (hops) {
                        COMM_TRY(comm->sendrecv_wait(comm->ctx, hop, stream), "sendrecv_wait");
                        float *sent = held[done];
                        held[done] = nullptr;
                        held[nxt] = spare;
                        spare = sent;
                    }
                }
                trained += n_pairs;
            }
        }
        lr *= tp->lr_decay;
    }

    // ---- the result: every rank receives both tables in node order
    float *stage = nullptr;
    if (buf.alloc(&stage, std::max<uint64_t>(gn2v::stripe_count(n, 0, world), max_part_rows) * ld * 4))
        return 1;
    for (uint32_t r = 0; r < world; ++r) {
        const uint64_t rows = gn2v::stripe_count(n, r, world);
        if (r == rank)
            HIP_TRY(hipMemcpyAsync(stage, central, rows * ld * 4, hipMemcpyDeviceToDevice, s));
        if (world > 1)
            COMM_TRY(comm->broadcast(comm->ctx, stage, rows * ld * 4, r, stream), "broadcast");
        if (scatter_rows(d_central, stage, rows, ld, r, world, s)) return 1;
    }
    {
        // who holds which part now (the rotation stops anywhere): every rank learns it
        std::vector<uint64_t> ids;
        for (uint32_t p = 0; p < parts; ++p)
            if (held[p]) ids.push_back(p);
        if (ids.size() != per_rank) return fail("gn2v_train_world: lost track of a context part");
        uint64_t *d_ids = nullptr, *d_all = nullptr;
        if (buf.alloc(&d_ids, per_rank * 8) || buf.alloc(&d_all, (size_t)parts * 8)) return 1;
        HIP_TRY(hipMemcpyAsync(d_ids, ids.data(), per_rank * 8, hipMemcpyHostToDevice, s));
        if (world > 1)
            COMM_TRY(comm->all_gather(comm->ctx, d_ids, d_all, per_rank * 8, stream), "all_gather");
        else
            HIP_TRY(hipMemcpyAsync(d_all, d_ids, per_rank * 8, hipMemcpyDeviceToDevice, s));
        std::vector<uint64_t> owners(parts);
        HIP_TRY(hipMemcpyAsync(owners.data(), d_all, (size_t)parts * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < parts; ++i) {
            const uint32_t r = i / per_rank, p = (uint32_t)owners[i];
            if (p >= parts) return fail("gn2v_train_world: the ranks disagree about the parts");
            const uint64_t rows = gn2v::stripe_count(n, p, parts);
            if (r == rank)
                HIP_TRY(hipMemcpyAsync(stage, held[p], rows * ld * 4, hipMemcpyDeviceToDevice, s));
            if (world > 1)
                COMM_TRY(comm->broadcast(comm->ctx, stage, rows * ld * 4, r, stream), "broadcast");
            if (scatter_rows(d_contextual, stage, rows, ld, p, parts, s)) return 1;
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (stats) {
        if (gn2v_stats_read(g, stats, s)) return 1;
        stats->block_parts = parts;
        stats->block_slices = plan.slices;
        stats->block_stripes = 1;
        stats->block_group_parts = group_parts;
        stats->block_round_walks = round_walks;
    }
    (void)trained;
    buf.done = true;
    return 0;
}

/* Drives every oracle entry point on a small synthetic graph; built with
 * -fsanitize=address,undefined by `make sanitize` (GPU sanitizers are unavailable on the pool, so
 * memory-safety checking happens on this CPU restatement).  TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gn2v_oracle.c"
#define GN2V_ORACLE_INLINE
#include "gn2v_cpu.c"

int main(void) {
    /* ring of 6 cliques of 5 nodes, plus one isolated node and one directed trap */
    enum { NC = 6, CS = 5, N = NC * CS + 2 };
    uint32_t adj[N][N];
    memset(adj, 0, sizeof(adj));
    for (int c = 0; c < NC; ++c) {
        for (int i = 0; i < CS; ++i)
            for (int j = 0; j < CS; ++j)
                if (i != j) adj[c * CS + i][c * CS + j] = 1;
        int a = c * CS, b = ((c + 1) % NC) * CS + 1;
        adj[a][b] = adj[b][a] = 1;
    }
    adj[0][N - 1] = 1; /* N-1 is a trap (no out edges); N-2 is isolated */
    uint64_t row_ptr[N + 1];
    uint32_t col[N * N];
    float cumw[N * N];
    uint32_t sources[N];
    uint64_t e = 0, ns = 0;
    for (int u = 0; u < N; ++u) {
        row_ptr[u] = e;
        float acc = 0.f;
        for (int v = 0; v < N; ++v)
            if (adj[u][v]) {
                col[e] = (uint32_t)v;
                acc += 1.0f + (float)((u + v) % 3);
                cumw[e++] = acc;
            }
        if (e > row_ptr[u]) sources[ns++] = (uint32_t)u;
    }
    row_ptr[N] = e;
    o_graph g = {N, e, row_ptr, col, NULL};
    o_graph gw = {N, e, row_ptr, col, cumw};

    const float weights[4][2] = {{1.f, 1.f}, {0.25f, 4.f}, {2.f, 0.5f}, {1.f, 1e-4f}};
    uint32_t *walks = malloc(sizeof(uint32_t) * ns * 3 * 20);
    for (int w = 0; w < 4; ++w) {
        o_walk_params wp = {20, 3, weights[w][0], weights[w][1], 100, 0};
        o_walks(&g, &wp, sources, ns, 7, w, 0, ns * 3, walks);
        o_walks(&gw, &wp, sources, ns, 7, w, 0, ns * 3, walks);
    }
    /* typed transitions, including weights small enough to reach the exact scan */
    uint32_t ntype[N], etype[N * N];
    for (int u = 0; u < N; ++u) ntype[u] = u % 5 == 4 ? 0xFFFFFFFFu : (uint32_t)(u % 3);
    for (uint64_t i = 0; i < e; ++i) etype[i] = (uint32_t)((i * 7) % 4);
    o_graph gt = {N, e, row_ptr, col, NULL, ntype, etype};
    o_graph gtw = {N, e, row_ptr, col, cumw, ntype, etype};
    const float tweights[3][2] = {{3.f, 0.25f}, {1e-5f, 1.f}, {1.f, 1e-5f}};
    for (int w = 0; w < 3; ++w) {
        o_walk_params wp = {20, 3, 0.5f, 2.f, 100, 0, tweights[w][0], tweights[w][1]};
        o_walks(&gt, &wp, sources, ns, 9, w, 0, ns * 3, walks);
        o_walks(&gtw, &wp, sources, ns, 9, w, 0, ns * 3, walks);
    }
    {
        o_walk_params wp = {20, 3, 0.25f, 4.f, 100, 0};
        o_walks(&g, &wp, sources, ns, 7, 1, 0, ns * 3, walks); /* walks used below */
    }
    int32_t *ctx = malloc(sizeof(int32_t) * ns * 3 * 20 * 6), *words = malloc(sizeof(int32_t) * ns * 3 * 20);
    o_window_batch(walks, ns * 3, 20, 3, ctx, words);
    uint32_t *pairs = malloc(sizeof(uint32_t) * ns * 3 * 20 * 6 * 2);
    uint64_t np = o_walk_pairs(walks, ns * 3, 20, 3, 1, pairs);

    float *c = malloc(sizeof(float) * N * 12), *x = malloc(sizeof(float) * N * 12);
    for (uint32_t model = 0; model < 2; ++model)
        for (uint32_t flags = 0; flags < 8; ++flags)
            for (uint32_t md = 1; md <= 3; md += 2) {
                o_walk_params wp = {20, 2, 0.5f, 2.0f, 100, 0};
                o_train_params tp = {model, 10, 12, 2, 4, 3, 0.02f, 0.9f, 6.0f, flags, 0.3f, md};
                o_fit(&g, &wp, &tp, sources, ns, 11, c, x, flags & 1 ? 2 : 1);
            }
    /* the CPU twins of the boundary (gn2v_cpu.h): graph, walks, batch, steps, whole fits */
    {
        gn2v_cpu_graph *cg = NULL;
        if (gn2v_cpu_graph_create(row_ptr, col, NULL, sources, N, e, ns, 0, 2, &cg)) return 1;
        gn2v_walk_params wp = {20, 2, 0.5f, 2.0f, 4, 0, 0.f, 0.f};
        if (gn2v_cpu_walks(cg, &wp, 3, 0, 5, ns * 2, walks, NULL)) return 1;
        if (gn2v_cpu_window_batch(walks, ns * 2, 20, 3, ctx, words, NULL)) return 1;
        for (uint32_t model = 0; model < 2; ++model) {
            gn2v_train_params tp = {model, 10, 12, 2, 4, 3, 0.02f, 0.9f, 6.0f, 1u | 8u, 0.3f, 0};
            gn2v_stats st;
            if (gn2v_cpu_init_table(c, N, 10, 12, 3, 0, 0.3f, NULL)) return 1;
            if (gn2v_cpu_init_table(x, N, 10, 12, 3, 1, 0.3f, NULL)) return 1;
            if (gn2v_cpu_sgns_step(cg, &tp, walks, ns * 2, 20, 3, 0, 5, 0.02f, c, x, NULL, NULL)) return 1;
            if (gn2v_cpu_cbow_step(cg, &tp, walks, ns * 2, 20, 3, 0, 5, 0.02f, c, x, NULL, NULL)) return 1;
            if (gn2v_cpu_train(cg, &wp, &tp, 5, ns + 3, c, x, &st, NULL)) return 1;
            if (st.walk_steps == 0 || st.pairs == 0) return 1;
        }
        if (!gn2v_cpu_walks(cg, NULL, 3, 0, 5, 1, walks, NULL)) return 1; /* refused, with a message */
        gn2v_cpu_graph_destroy(cg);
    }
    /* two-node walks through the general step with a pool and row indirection */
    o_train_params tp = {0, 10, 12, 1, 4, 1, 0.02f, 0.9f, 6.0f, 1, 0.3f, 1};
    uint32_t *rows = malloc(sizeof(uint32_t) * np * 2);
    for (uint64_t i = 0; i < np * 2; ++i) rows[i] = pairs[i];
    o_step_io io;
    memset(&io, 0, sizeof(io));
    io.walks = pairs;
    io.walk_rows = rows;
    io.central = c;
    io.contextual = x;
    io.neg_pool = col;
    io.neg_pool_size = e;
    io.neg_id_mul = 1;
    o_train_walks_ex(&g, &tp, &io, np, 2, 3, 0, 0, 0.02f, 1);

    /* block-partitioned schedule: extraction, stable sort, alias tables, one round over every part */
    {
        o_block_plan bp = {3, 1, 6, 2, 20, 3, 1, 4, o_block_row_bits(N, 3), 0, 6, 2, 0,
                           o_block_ctx_bits(N, 6, 2)};
        bp.key_bits = o_block_cell_bits(6, 2) + bp.row_bits + bp.ctx_bits;
        uint64_t nw = ns * 3, cap = nw * 20 * 6;
        uint64_t *bk = malloc(sizeof(uint64_t) * cap);
        uint32_t hub[(N + 31) / 32];
        uint64_t *alias = malloc(sizeof(uint64_t) * N), poff[13];
        uint32_t *hot_list = malloc(sizeof(uint32_t) * 12 * 192);
        uint8_t *hot_slot = malloc(N);
        o_block_alias(&g, 6, 2, bp.hot_rows, alias, poff, hub, hot_list, hot_slot, NULL);
        free(hot_list);
        free(hot_slot);
        /* two groups of parts (the second wraps round), then the whole round */
        uint64_t n_a = o_block_extract(&g, &bp, walks, nw, 7, 1, 0, 4, 3, hub, bk, NULL);
        uint64_t n_b = o_block_extract(&g, &bp, walks, nw, 7, 1, 0, 1, 3, hub, bk, NULL);
        uint64_t nb = o_block_extract(&g, &bp, walks, nw, 7, 1, 0, 0, 0, hub, bk, NULL);
        if (n_a + n_b != nb) return 1;
        o_block_sort(bk, nb, bp.ctx_bits);
        uint64_t off[13];
        o_block_cell_offsets(bk, nb, bp.row_bits + bp.ctx_bits, 12, off);
        uint64_t crows = (N + 3 - 1 - 1) / 3 + 1, xrows = N / 6 + 1, trained = 0;
        float *bc = malloc(sizeof(float) * crows * 12), *bx = malloc(sizeof(float) * xrows * 12);
        o_init_table_rows(bc, (N - 1 + 2) / 3, 10, 12, 5, 0, 0.3f, 1, 3);
        for (uint32_t part = 0; part < 6; ++part) {
            o_init_table_rows(bx, (N - part + 5) / 6, 10, 12, 5, 1, 0.3f, part, 6);
            trained += o_block_step(&g, &tp, &bp, bk, off, alias, poff, bc, bx, 2, part, 7, 1,
                                    0.02f, NULL, 0);
        }
        if (trained != nb) return 1;
        /* the same round under a placement: inside the classes mod 6 (part buffers), then over
         * the whole graph (one natural table) */
        uint32_t *place = malloc(sizeof(uint32_t) * N), *inv = malloc(sizeof(uint32_t) * N);
        float *whole = malloc(sizeof(float) * N * 12);
        hot_list = malloc(sizeof(uint32_t) * 12 * 192);
        hot_slot = malloc(N);
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t classes = pass ? 1 : 6;
            o_block_placement(N, classes, 7, 3, place, inv);
            for (uint32_t x = 0; x < N; ++x)
                if (inv[place[x]] != x || place[x] % classes != x % classes) return 1;
            o_block_alias(&g, 6, 2, 0, alias, poff, hub, hot_list, hot_slot, inv);
            uint64_t np2 = o_block_extract(&g, &bp, walks, nw, 7, 1, 0, 0, 0, NULL, bk, place);
            if (np2 != nb) return 1;
            o_block_sort(bk, np2, bp.ctx_bits);
            o_block_cell_offsets(bk, np2, bp.row_bits + bp.ctx_bits, 12, off);
            o_init_table(whole, N, 10, 12, 5, 1, 0.3f);
            trained = 0;
            for (uint32_t part = 0; part < 6; ++part) {
                o_init_table_rows(bx, (N - part + 5) / 6, 10, 12, 5, 1, 0.3f, part, 6);
                trained += o_block_step(&g, &tp, &bp, bk, off, alias, poff, bc,
                                        classes == 1 ? whole : bx, 2, part, 7, 1, 0.02f, inv,
                                        classes == 1);
            }
            if (trained != nb) return 1;
        }
        free(place); free(inv); free(whole); free(hot_list); free(hot_slot);
        free(bk); free(alias); free(bc); free(bx);
    }

    uint32_t *bs = malloc(sizeof(uint32_t) * 999 * 4), *bd = malloc(sizeof(uint32_t) * 999 * 4);
    o_ba_edges(1000, 4, 42, bs, bd);
    double acc = 0;
    for (int i = 0; i < N * 12; ++i) acc += c[i] + x[i];
    printf("ok %llu pairs, checksum %.6f\n", (unsigned long long)np, acc);
    free(walks); free(ctx); free(words); free(pairs); free(c); free(x); free(rows); free(bs); free(bd);
    return 0;
}

/* The whole hot path from a plain C program: no Python, no PyTorch in the process -- what a
 * compiled-language binding of embiggen's one call (embedders/ensmallen_embedders/node2vec.py:99)
 * would do.  A ring of cliques (8 cliques of 8 nodes) is embedded with Node2Vec SkipGram and with
 * CBOW through gn2v_train into tables from hipMalloc; nodes of one clique must end closer
 * (cosine of the central vectors) than nodes of different cliques.  Built with hipcc and run on
 * the GPU by tests/test_gpu_integration_doc.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gn2v.h"

#define CLIQUES 8
#define SIZE 8
#define N (CLIQUES * SIZE)
#define D 16

static int die(const char *what) {
    fprintf(stderr, "%s: %s\n", what, gn2v_last_error());
    return 1;
}

static double cosine(const float *a, const float *b) {
    double ab = 0, aa = 0, bb = 0;
    for (int i = 0; i < D; ++i) {
        ab += (double)a[i] * b[i];
        aa += (double)a[i] * a[i];
        bb += (double)b[i] * b[i];
    }
    return ab / (sqrt(aa * bb) + 1e-30);
}

int main(void) {
    if (gn2v_device_count() < 1) return die("no HIP device");
    /* CSR, neighbours ascending: the clique mates plus one bridge to the next / previous clique */
    static uint64_t row_ptr[N + 1];
    static uint32_t col_idx[N * (SIZE + 1)];
    uint64_t e = 0;
    for (uint32_t v = 0; v < N; ++v) {
        uint32_t c = v / SIZE, i = v % SIZE, nb[SIZE + 1], m = 0;
        for (uint32_t j = 0; j < SIZE; ++j)
            if (j != i) nb[m++] = c * SIZE + j;
        if (i == SIZE - 1) nb[m++] = ((c + 1) % CLIQUES) * SIZE;            /* bridge out */
        if (i == 0) nb[m++] = ((c + CLIQUES - 1) % CLIQUES) * SIZE + SIZE - 1; /* bridge in  */
        for (uint32_t a = 0; a < m; ++a) /* insertion sort */
            for (uint32_t b = a + 1; b < m; ++b)
                if (nb[b] < nb[a]) {
                    uint32_t t = nb[a];
                    nb[a] = nb[b];
                    nb[b] = t;
                }
        row_ptr[v] = e;
        for (uint32_t a = 0; a < m; ++a) col_idx[e++] = nb[a];
    }
    row_ptr[N] = e;

    gn2v_graph *g = NULL;
    if (gn2v_graph_create(row_ptr, col_idx, NULL, NULL, N, e, N, GN2V_GRAPH_SYMMETRIC, 0, &g))
        return die("gn2v_graph_create");

    float *d_central = NULL, *d_contextual = NULL;
    if (hipMalloc((void **)&d_central, sizeof(float) * N * D) != hipSuccess ||
        hipMalloc((void **)&d_contextual, sizeof(float) * N * D) != hipSuccess)
        return die("hipMalloc");

    for (uint32_t model = GN2V_MODEL_SKIPGRAM; model <= GN2V_MODEL_CBOW; ++model) {
        gn2v_walk_params wp;
        memset(&wp, 0, sizeof wp);
        wp.walk_length = 32;
        wp.iterations = 10;
        wp.return_weight = 1.0f;
        wp.explore_weight = 1.0f;
        wp.max_neighbours = 100;
        gn2v_train_params tp;
        memset(&tp, 0, sizeof tp);
        tp.model = model;
        tp.d = D;
        tp.ld = D;
        tp.epochs = 10;
        tp.k = 5;
        tp.window = 4;
        tp.lr = 0.05f;
        tp.lr_decay = 0.9f;
        tp.clip = 6.0f;
        tp.flags = GN2V_TRAIN_SCALE_FREE;
        tp.init_scale = 0.25f;
        gn2v_stats st;
        memset(&st, 0, sizeof st);
        /* the handle's counters accumulate over calls until they are reset */
        if (gn2v_stats_reset(g, NULL)) return die("gn2v_stats_reset");
        if (gn2v_train(g, &wp, &tp, 42, 0, d_central, d_contextual, &st, NULL))
            return die("gn2v_train");
        static float central[N * D];
        if (hipMemcpy(central, d_central, sizeof central, hipMemcpyDeviceToHost) != hipSuccess)
            return die("hipMemcpy");
        double same = 0, other = 0;
        uint32_t n_same = 0, n_other = 0;
        for (uint32_t a = 0; a < N; ++a)
            for (uint32_t b = a + 1; b < N; ++b) {
                double c = cosine(central + a * D, central + b * D);
                if (!isfinite(c)) return die("non-finite embedding");
                if (a / SIZE == b / SIZE) {
                    same += c;
                    ++n_same;
                } else {
                    other += c;
                    ++n_other;
                }
            }
        same /= n_same;
        other /= n_other;
        printf("model %u: pairs %llu walk_steps %llu centres %llu, cosine same clique %.3f, "
               "other cliques %.3f\n",
               model, (unsigned long long)st.pairs, (unsigned long long)st.walk_steps,
               (unsigned long long)st.centres, same, other);
        if (st.walk_steps != 10ull * 10 * N * 31) return die("unexpected number of walk steps");
        if (!(same > other + 0.3)) return die("cliques not separated");
    }
    (void)hipFree(d_central);
    (void)hipFree(d_contextual);
    if (gn2v_graph_destroy(g)) return die("gn2v_graph_destroy");
    puts("ok");
    return 0;
}

/* The multi-GPU fit from a plain C program: gn2v_train_world with the communicator of
 * include/gn2v_rccl.h -- RCCL, no Python, no PyTorch in the process.  One process per GPU:
 *
 *     fit_world_rccl <rank> <world> <id file>
 *
 * Rank 0 makes the job's id (gn2v_rccl_unique_id) and writes it to <id file>; the other ranks
 * wait for the file.  Every rank embeds the same ring of cliques and must receive the same two
 * tables, in which the nodes of a clique are closer than nodes of different cliques.  On the
 * one-GPU test box tests/test_gpu_integration_doc.py runs it with world = 1 (RCCL initialised,
 * the agreement all-gather through it); the N > 1 form is what a launcher starts once per GPU. */
#define _DEFAULT_SOURCE /* usleep */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gn2v.h"
#include "gn2v_rccl.h"

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

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <rank> <world> <id file>\n", argv[0]);
        return 2;
    }
    const uint32_t rank = (uint32_t)atoi(argv[1]), world = (uint32_t)atoi(argv[2]);
    const char *id_file = argv[3];
    const int devices = gn2v_device_count();
    if (devices < 1) return die("no HIP device");
    const int device = (int)(rank % (uint32_t)devices);

    /* the job's id: rank 0 makes it, the others read it once it is complete */
    unsigned char id[GN2V_RCCL_ID_BYTES];
    if (rank == 0) {
        if (gn2v_rccl_unique_id(id)) return die("gn2v_rccl_unique_id");
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) || rename(tmp, id_file)) {
            fprintf(stderr, "cannot write %s\n", id_file);
            return 1;
        }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); ++tries) usleep(100000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) {
            fprintf(stderr, "cannot read %s\n", id_file);
            return 1;
        }
        fclose(f);
    }
    gn2v_comm comm;
    memset(&comm, 0, sizeof comm);
    if (gn2v_rccl_comm_create(id, rank, world, device, &comm)) return die("gn2v_rccl_comm_create");

    /* CSR, neighbours ascending: the clique mates plus one bridge to the next / previous clique */
    static uint64_t row_ptr[N + 1];
    static uint32_t col_idx[N * (SIZE + 1)];
    uint64_t e = 0;
    for (uint32_t v = 0; v < N; ++v) {
        uint32_t c = v / SIZE, i = v % SIZE, nb[SIZE + 1], m = 0;
        for (uint32_t j = 0; j < SIZE; ++j)
            if (j != i) nb[m++] = c * SIZE + j;
        if (i == SIZE - 1) nb[m++] = ((c + 1) % CLIQUES) * SIZE;
        if (i == 0) nb[m++] = ((c + CLIQUES - 1) % CLIQUES) * SIZE + SIZE - 1;
        for (uint32_t a = 0; a < m; ++a)
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

    if (hipSetDevice(device) != hipSuccess) return die("hipSetDevice");
    gn2v_graph *g = NULL;
    if (gn2v_graph_create(row_ptr, col_idx, NULL, NULL, N, e, N, GN2V_GRAPH_SYMMETRIC, device, &g))
        return die("gn2v_graph_create");
    float *d_central = NULL, *d_contextual = NULL;
    if (hipMalloc((void **)&d_central, sizeof(float) * N * D) != hipSuccess ||
        hipMalloc((void **)&d_contextual, sizeof(float) * N * D) != hipSuccess)
        return die("hipMalloc");

    gn2v_walk_params wp;
    memset(&wp, 0, sizeof wp);
    wp.walk_length = 32;
    wp.iterations = 10;
    wp.return_weight = 1.0f;
    wp.explore_weight = 1.0f;
    wp.max_neighbours = 100;
    gn2v_train_params tp;
    memset(&tp, 0, sizeof tp);
    tp.model = GN2V_MODEL_SKIPGRAM;
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
    if (gn2v_stats_reset(g, NULL)) return die("gn2v_stats_reset");
    if (gn2v_train_world(g, &wp, &tp, 42, 0, 0, &comm, d_central, d_contextual, &st, NULL))
        return die("gn2v_train_world");

    static float central[N * D];
    if (hipMemcpy(central, d_central, sizeof central, hipMemcpyDeviceToHost) != hipSuccess)
        return die("hipMemcpy");
    double same = 0, other = 0, sum = 0;
    uint32_t n_same = 0, n_other = 0;
    for (uint32_t a = 0; a < N * D; ++a) sum += central[a];
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
    /* every rank prints the same checksum: all of them received the same tables */
    printf("rank %u of %u: pairs %llu (this rank), parts %u, cosine same clique %.3f, other "
           "cliques %.3f, checksum %.6f\n",
           rank, world, (unsigned long long)st.pairs, st.block_parts, same, other, sum);
    if (st.pairs == 0) return die("no pair trained");
    if (!(same > other + 0.3)) return die("cliques not separated");
    (void)hipFree(d_central);
    (void)hipFree(d_contextual);
    if (gn2v_graph_destroy(g)) return die("gn2v_graph_destroy");
    if (gn2v_rccl_comm_destroy(&comm)) return die("gn2v_rccl_comm_destroy");
    if (rank == 0) remove(id_file);
    puts("ok");
    return 0;
}

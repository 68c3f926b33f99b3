/* A plain C11 caller of the drop-in boundary (what a cgo / Rust FFI / JNI binding links against):
 * include/gn2v.h must be valid C and the host-only entry points must work without a GPU.
 * Built and run by tests/test_cabi.py. */
#include <stdio.h>
#include <string.h>

#include "gn2v_internal.h" /* the host-only planning entry points live there */

int main(void) {
    if (gn2v_version() < 200) return 1;
    uint32_t parts = 0, slices = 0, group = 0;
    if (gn2v_block_auto_plan(10000000, 8, 128, 10, &parts, &slices) || parts != 16 || slices != 2841) return 2;
    if (gn2v_block_auto_plan(10000000, 8, 0, 10, &parts, &slices) || parts != 32 || slices != 8) return 2;
    uint64_t walks = 0;
    if (gn2v_block_round_plan(270000000000ull, 10000000, 128, 5, 1, 38, 8, 0, &walks, &group) ||
        walks != (1u << 23) || group != 10)
        return 3;
    if (gn2v_block_round_plan(1, 10000000, 1, 5, 1, 38, 8, 0, &walks, &group) == 0)
        return 4; /* a walk of one node */
    if (strlen(gn2v_last_error()) == 0) return 5;
    gn2v_graph *g = NULL;
    if (gn2v_graph_create(NULL, NULL, NULL, NULL, 2, 2, 2, 0, 0, &g) == 0) return 6;
    if (strstr(gn2v_last_error(), "NULL") == NULL) return 7;
    gn2v_block_plan plan;
    memset(&plan, 0, sizeof plan);
    printf("gn2v %u: sizeof(gn2v_block_plan) = %zu, sizeof(gn2v_stats) = %zu, "
           "sizeof(gn2v_block_io) = %zu\n",
           gn2v_version(), sizeof plan, sizeof(gn2v_stats), sizeof(gn2v_block_io));
    return 0;
}

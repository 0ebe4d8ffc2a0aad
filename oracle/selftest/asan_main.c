/* Sanitizer self-test of the oracle (CPU build only): scalar vs striped on random and planted pairs, plus the
 * annotate logic on hand-made records.  Build + run: make -C oracle asan   (gcc -fsanitize=address,undefined) */
#include "../fade_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static unsigned long long s = 88172645463325252ULL;
static unsigned rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(void) {
    fo_params p;
    fo_params_default(&p);
    const char *al = "ACGTNRY";
    int bad = 0;
    for (int it = 0; it < 3000; it++) {
        int lq = 1 + rnd() % 300, lr = 1 + rnd() % 900;
        char *q = malloc(lq + 1), *r = malloc(lr + 1);
        for (int i = 0; i < lq; i++) q[i] = al[rnd() % (it % 3 ? 4 : 7)];
        for (int i = 0; i < lr; i++) r[i] = al[rnd() % (it % 3 ? 4 : 7)];
        if (it % 2 && lq > 10 && lr > lq) memcpy(r + rnd() % (lr - lq), q + lq / 2, lq / 2);
        fo_sw_result a, b;
        uint32_t oa[16], ob[16];
        fo_sw_trace(&p, q, lq, r, lr, &a, oa, 16);
        fo_sw_striped(&p, q, lq, r, lr, &b, ob, 16);
        int n = a.n_ops < 16 ? a.n_ops : 16;
        if (memcmp(&a, &b, sizeof a) || memcmp(oa, ob, 4 * n)) bad++;
        free(q);
        free(r);
    }
    /* annotate logic on a read with both clips near a contig end */
    const char *names[1] = {"c1"};
    char *ref = malloc(2001);
    for (int i = 0; i < 2000; i++) ref[i] = "ACGT"[rnd() % 4];
    ref[2000] = 0;
    const char *seqs[1] = {ref};
    int64_t lens[1] = {2000};
    fo_genome g = {1, names, lens, seqs};
    uint32_t cig[3] = {(20u << 4) | 4, (110u << 4) | 0, (20u << 4) | 4};
    uint8_t seq4[75], qual[150];
    for (int i = 0; i < 75; i++) seq4[i] = (uint8_t)(((1u << (rnd() % 4)) << 4) | (1u << (rnd() % 4)));
    memset(qual, 30, sizeof qual);
    for (int64_t pos = 0; pos < 2000; pos += 97) {
        fo_read rd = {"r", 0, 0, pos, 3, cig, 150, seq4, qual, 1};
        fo_anno an;
        fo_annotate_task(&p, &g, &rd, 5, 300, &an);
        fo_anno_free(&an);
    }
    free(ref);
    printf("oracle sanitizer self-test: scalar/striped mismatches %d\n", bad);
    return bad != 0;
}

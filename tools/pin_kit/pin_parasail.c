/* pin_parasail.c — NOT COMPILED OR RUN IN THE BUILD ENVIRONMENT (no libparasail there).  Written against the public
 * parasail C API as recollected (parasail.h of 2.4.x): parasail_matrix_create, parasail_sw_trace_striped_16,
 * parasail_result_get_{score,end_query,end_ref}, parasail_result_get_cigar, parasail_cigar_decode_{op,len}.
 *
 * For every (query, ref) pair of tests/golden/sw_pairs.tsv it runs what dparasail's Parasail("ACTGN", 10, 2, 2, -3)
 * .sw_striped(q, r) runs (source/anno.d:36, source/analysis.d:67) and prints the line in the golden file's format:
 *   query ref score end_query end_ref beg_query position n_ops cigar_front16
 * The CIGAR is printed as dparasail hands it to FADE ACCORDING TO SURVEY Appendix A.6: parasail's ops, padded with S for
 * the unaligned query ends.  The raw parasail CIGAR follows as a 10th column so that the padding rule itself can be checked
 * against pin_dparasail.d's output.
 *   gcc -O2 -o pin_parasail pin_parasail.c -lparasail && ./pin_parasail tests/golden/sw_pairs.tsv
 */
#include <parasail.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s sw_pairs.tsv\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 1; }
    parasail_matrix_t *m = parasail_matrix_create("ACTGN", 2, -3);
    static char line[1 << 20];
    printf("#query\tref\tscore\tend_query\tend_ref\tbeg_query\tposition\tn_ops\tcigar_front16\traw_parasail_cigar\n");
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        char *q = strtok(line, "\t\n"), *r = strtok(NULL, "\t\n");
        if (!q || !r) continue;
        const int ql = (int)strlen(q), rl = (int)strlen(r);
        parasail_result_t *res = parasail_sw_trace_striped_16(q, ql, r, rl, 10, 2, m);
        const int score = parasail_result_get_score(res);
        const int end_q = parasail_result_get_end_query(res), end_r = parasail_result_get_end_ref(res);
        parasail_cigar_t *c = parasail_result_get_cigar(res, q, ql, r, rl, m);
        char raw[8192] = "", padded[8192] = "";
        int n_ops = 0, at = 0, shown = 0;
        if (c->beg_query > 0) { at += sprintf(padded + at, "%dS", c->beg_query); n_ops++; shown++; }
        for (int i = 0; i < c->len; i++) {
            const char op = parasail_cigar_decode_op(c->seq[i]);
            const uint32_t len = parasail_cigar_decode_len(c->seq[i]);
            sprintf(raw + strlen(raw), "%u%c", len, op);
            if (shown < 16) { at += sprintf(padded + at, "%u%c", len, op); shown++; }
            n_ops++;
        }
        if (ql - 1 - end_q > 0) { if (shown < 16) sprintf(padded + at, "%dS", ql - 1 - end_q); n_ops++; }
        printf("%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%s\t%s\n", q, r, score, end_q, end_r, c->beg_query, c->beg_ref, n_ops, padded, raw);
        parasail_cigar_free(c);
        parasail_result_free(res);
    }
    parasail_matrix_free(m);
    fclose(f);
    return 0;
}

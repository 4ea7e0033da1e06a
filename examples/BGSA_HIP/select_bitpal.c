/* Linked into the BitPAl build of the drop-in example: the one line a BGSA_HIP main() adds after
 * handle_args(), done here as a constructor so that the reference's main.c stays
 * untouched.  Scores other than 2/-3/-5: BGSA_HIP_SCORES="1,-3,-2" in the environment. */
#include <stdio.h>
#include <stdlib.h>

int bgsa_hip_select_scores(int match, int mismatch, int gap);
const char *bgsa_hip_last_error(void);

__attribute__((constructor)) static void select_bitpal(void)
{
    int m = 2, x = -3, g = -5;
    const char *e = getenv("BGSA_HIP_SCORES");
    if (e && sscanf(e, "%d,%d,%d", &m, &x, &g) != 3) {
        fprintf(stderr, "BGSA_HIP_SCORES must be match,mismatch,gap\n");
        exit(1);
    }
    if (bgsa_hip_select_scores(m, x, g) != 0) {
        fprintf(stderr, "%s\n", bgsa_hip_last_error());
        exit(1);
    }
}

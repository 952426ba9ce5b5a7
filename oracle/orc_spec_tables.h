/* oracle/orc_spec_tables.h — TEST INFRASTRUCTURE ONLY.
 *
 * A/52 log-addition table ("latab", Table 7.14 of the standard; the reference
 * carries it as liba52/bit_allocate.c:78-101 (negated, int8) and as
 * src/ac3enc/ac3tab.h:51-80).  The table is non-increasing from 64 down to 0,
 * so it is kept here as run lengths: la_run[k] = how many consecutive entries
 * hold the value 64-k.
 */
#ifndef ORC_SPEC_TABLES_H
#define ORC_SPEC_TABLES_H
#include <stdint.h>

static const uint8_t la_run[65] = {
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1,
    1, 2, 1, 1, 2, 1, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2,
    2, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 3, 3, 3,
    3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 8, 8, 11, 14, 19, 32,
    46
};

static inline void orc_build_logadd(uint8_t *tab256)
{
    int n = 0;
    for (int k = 0; k <= 64; k++)
        for (int r = 0; r < la_run[k]; r++) tab256[n++] = (uint8_t)(64 - k);
}
#endif

/* oracle/ref_ac3tab_glue.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Our own accessor, compiled against the reference's UNMODIFIED encoder headers where they lie
 * (-I/root/reference/src/ac3enc: common.h + ac3tab.h, no stand-in header needed), into
 * oracle/_ref/ac3tab_ref.so.  It hands out the constant tables of the encoder so that
 * tests/golden/make_golden.py can freeze them in tests/golden/ac3tab.npz; the CPU gate then checks
 * the encoder oracle's and the engine's tables against that fixture (tests/test_oracle_golden.py).
 *
 * Only the spec tables of ac3tab.h:3-171 are pinned this way.  The runtime tables (costab, sintab,
 * xcos1, xsin1, fft_rev, crc_table, bndtab, masktab) are filled by code in ac3enc.cpp, which needs
 * <windows.h>/<crtdbg.h> and is not built here.
 */
#include <string.h>
#include "common.h"
#include "ac3tab.h"

extern "C" const void *refglue_ac3tab(const char *name, int *count, int *elem_bytes)
{
#define TAB(t) if (!strcmp(name, #t)) { *count = (int)(sizeof(t) / sizeof(t[0])); *elem_bytes = (int)sizeof(t[0]); return (const void *)(t); }
#define TAB2(t) if (!strcmp(name, #t)) { *count = (int)(sizeof(t) / sizeof(t[0][0])); *elem_bytes = (int)sizeof(t[0][0]); return (const void *)(t); }
    TAB(ac3_freqs) TAB(ac3_bitratetab) TAB(ac3_window) TAB(latab) TAB2(hth) TAB(baptab)
    TAB(sdecaytab) TAB(fdecaytab) TAB(sgaintab) TAB(dbkneetab) TAB(floortab) TAB(fgaintab) TAB(bndsz)
#undef TAB
#undef TAB2
    *count = 0;
    *elem_bytes = 0;
    return 0;
}

/* oracle/ref_a52dec_glue.c — TEST INFRASTRUCTURE.  A minimal elementary-stream front end for the REAL liba52 and
 * the REAL libao file drivers (audio_out_wav.c, audio_out_float.c, audio_out_null.c, convert2s16.c), all compiled
 * from the reference's own sources where they lie (see Makefile: target _ref/a52dec_ref).  The reference's own
 * front end, src/a52dec.c, cannot be built here: its vc++/config.h selects <io.h> and the Win32 audio driver.
 * This file only feeds frames to a52_frame/a52_block and hands the planes to the reference's output drivers, in
 * the order a52dec.c:240-310 does; tests/test_tools_gpu.py compares tools/ac3mi_dec against its output.
 *
 *   a52dec_ref <mode> <disable_dynrng> <disable_adjust> <gain_dB> <file>   > out
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "a52.h"
#include "audio_out.h"

extern ao_open_t ao_wav_open, ao_wavdolby_open, ao_wav6_open, ao_float_open, ao_null_open, ao_null4_open, ao_null6_open;

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    ao_open_t *open = NULL;
    if (!strcmp(argv[1], "wav")) open = ao_wav_open;
    else if (!strcmp(argv[1], "wavdolby")) open = ao_wavdolby_open;
    else if (!strcmp(argv[1], "wav6")) open = ao_wav6_open;
    else if (!strcmp(argv[1], "float")) open = ao_float_open;
    else if (!strcmp(argv[1], "null")) open = ao_null_open;
    else if (!strcmp(argv[1], "null4")) open = ao_null4_open;
    else if (!strcmp(argv[1], "null6")) open = ao_null6_open;
    if (!open) return 2;
    const int disable_dynrng = atoi(argv[2]), disable_adjust = atoi(argv[3]);
    const double gain = pow(2, atof(argv[4]) / 6);
    FILE *in = fopen(argv[5], "rb");
    if (!in) return 3;
    ao_instance_t *output = open();
    a52_state_t *state = a52_init(0);
    static uint8_t buf[3840 + 64];
    int frames = 0, errors = 0;
    for (;;) {
        int flags, sample_rate, bit_rate;
        if (fread(buf, 1, 7, in) != 7) break;
        int length = a52_syncinfo(buf, &flags, &sample_rate, &bit_rate);
        while (!length) {                               /* resync: drop one byte */
            memmove(buf, buf + 1, 6);
            if (fread(buf + 6, 1, 1, in) != 1) goto done;
            length = a52_syncinfo(buf, &flags, &sample_rate, &bit_rate);
        }
        if ((int)fread(buf + 7, 1, (size_t)(length - 7), in) != length - 7) break;
        level_t level;
        sample_t bias;
        int ok = 0, i;
        do {
            if (output->setup(output, sample_rate, &flags, &level, &bias)) break;
            if (!disable_adjust) flags |= A52_ADJUST_LEVEL;
            level = (level_t)(level * gain);
            if (a52_frame(state, buf, &flags, &level, bias)) break;
            if (disable_dynrng) a52_dynrng(state, NULL, NULL);
            for (i = 0; i < 6; i++) {
                if (a52_block(state)) break;
                if (output->play(output, flags, a52_samples(state))) break;
            }
            ok = i == 6;
        } while (0);
        frames++;
        errors += !ok;
    }
done:
    output->close(output);
    a52_free(state);
    fprintf(stderr, "frames %d errors %d\n", frames, errors);
    return 0;
}

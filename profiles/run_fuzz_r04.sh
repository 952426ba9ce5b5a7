#!/bin/bash
# Round-4 randomised campaign: gpurun -- 'bash profiles/run_fuzz_r04.sh PART'  ->  gpurun_out/fuzz_r04_PART.txt (copied to
# profiles/r04_fuzz_campaign.txt).  New seeds.  PART enc: the rewritten mantissa packer under both packers (one wavefront per frame /
# per audio block), the transcoder (second-generation content: the out-of-contract quantiser path) and the byte-stream layer;
# PART dec: the decoder scripts under the front ends that are left (1 = the one-kernel reference, 4 / 5 = split, auto);
# PART end: the end-of-round campaign after the search's tighter bounds and the fused mantissa + transform kernel (new seeds).
PART=${1:-enc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fuzz_r04_$PART.txt
mkdir -p $R/gpurun_out; : > $OUT
t() { name=$1; shift; "$@" > /tmp/fz_$name.log 2>&1; echo "$name rc=$? $(tail -1 /tmp/fz_$name.log)" | tee -a $OUT; }
cd $R
if [ $PART = enc ]; then
AC3MI_ENCODE_MODE=1 t encode_m1 python tests/fuzz_encode.py 500 5101
AC3MI_ENCODE_MODE=2 t encode_m2 python tests/fuzz_encode.py 500 5102
t encode_auto python tests/fuzz_encode.py 200 5103
t transcode_auto python tests/fuzz_transcode.py 200 5104
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=1 t transcode_m4 python tests/fuzz_transcode.py 200 5105
t stream_auto python tests/fuzz_stream.py 150 5106
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=1 t stream_m4 python tests/fuzz_stream.py 100 5107
elif [ $PART = end ]; then
AC3MI_ENCODE_MODE=1 t encode_m1 python tests/fuzz_encode.py 1200 5301
AC3MI_ENCODE_MODE=2 t encode_m2 python tests/fuzz_encode.py 1200 5302
t transcode_auto python tests/fuzz_transcode.py 500 5303
AC3MI_DECODE_MODE=4 AC3MI_ENCODE_MODE=1 t transcode_m4 python tests/fuzz_transcode.py 300 5304
t mantx python tests/fuzz_mantx.py 400 5305
t pcm_auto python tests/fuzz_pcm.py 200 5306
AC3MI_DECODE_MODE=6 t pcm_m6 python tests/fuzz_pcm.py 200 5307
t stream_auto python tests/fuzz_stream.py 100 5308
AC3MI_DECODE_MODE=6 t corrupt_m6 python tests/fuzz_corrupt.py 60 5309
else
AC3MI_DECODE_MODE=4 t decode_m4 python tests/fuzz_decode.py 100 5201
AC3MI_DECODE_MODE=5 t decode_m5 python tests/fuzz_decode.py 100 5202
AC3MI_DECODE_MODE=1 t decode_m1 python tests/fuzz_decode.py 60 5203
t decode_auto python tests/fuzz_decode.py 100 5204
AC3MI_DECODE_MODE=4 t pcm_m4 python tests/fuzz_pcm.py 200 5205
t pcm_auto python tests/fuzz_pcm.py 200 5206
AC3MI_DECODE_MODE=4 t mix_m4 python tests/fuzz_mixlevel.py 100 5207
AC3MI_DECODE_MODE=4 t corrupt_m4 python tests/fuzz_corrupt.py 60 5208
AC3MI_DECODE_MODE=5 t corrupt_m5 python tests/fuzz_corrupt.py 40 5209
fi

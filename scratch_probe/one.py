import sys
sys.path.insert(0, '.')
import numpy as np
from tests import _harness as H
from tests.test_encode_gpu import _gpu, _oracle
pkg = H.pkg(); eng = pkg.Engine(0)
for nch, freq, br in ((1, 22050, 256000), (1, 48000, 256000)):
    pcm = [H.gen_pcm(4, nch, seed=77 + s, kind=("tones", "noise", "music", "bursts")[s]) for s in range(4)]
    want, wt = _oracle(pcm, nch, br, freq, tuple(range(8)))
    got, gt = _gpu(eng, pcm, nch, br, freq, tuple(range(8)), taps=True)
    bad = [(s, f) for s in range(4) for f in range(4) if not np.array_equal(got[s, f], want[s, f])]
    print(nch, freq, br, "bad frames", bad, "snr gpu", gt["snroffst"].reshape(4, 4, 2)[:, :, :].tolist(), "orc", wt["snr"].tolist())

// tests/dropin_c/enc_host.cpp - a C++ host that declares the encoder exactly as the reference's src/ac3enc/ac3enc.h:6-7
// does (C++ linkage, no extern "C": src/AC3ACM.cpp:60 includes that header plainly) and calls it the way
// stream_convert_pcm does (src/AC3ACM.cpp:1762, 1940), linked against libac3mi.so's mangled exports.
//   enc_host <in.s16> <out.ac3> <freq> <bitrate> <channels>
#include <cstdio>
#include <cstdlib>
#include <vector>

int AC3_encode_init(int freq, int bitrate, int channels);
int AC3_encode_frame(unsigned char *dst, short *samples, unsigned char *chmap);

int main(int argc, char **argv)
{
    if (argc != 6) return 2;
    FILE *in = std::fopen(argv[1], "rb"), *out = std::fopen(argv[2], "wb");
    if (!in || !out) return 2;
    const int channels = std::atoi(argv[5]);
    const int fl = AC3_encode_init(std::atoi(argv[3]), std::atoi(argv[4]), channels);     // frame bytes, 0 = refused
    if (fl <= 0) { std::printf("init refused\n"); return 1; }
    unsigned char chmap[6] = {0, 2, 1, 4, 5, 3};                                            // WAVE order -> coded order, 5.1
    if (channels != 6) for (int i = 0; i < 6; i++) chmap[i] = (unsigned char)i;
    std::vector<short> pcm(1536 * channels);
    std::vector<unsigned char> frame(4096);
    int n = 0;
    while (std::fread(pcm.data(), 2, pcm.size(), in) == pcm.size()) {
        const int bytes = AC3_encode_frame(frame.data(), pcm.data(), chmap);
        if (bytes != fl) { std::printf("frame %d: %d bytes, expected %d\n", n, bytes, fl); return 1; }
        std::fwrite(frame.data(), 1, bytes, out);
        n++;
    }
    std::printf("frames %d bytes %d\n", n, fl);
    std::fclose(in);
    std::fclose(out);
    return 0;
}

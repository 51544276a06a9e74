// btrx_amd -- command-line driver, counterpart of the reference's apps/btrx for the two hot-path
// blocks.  Same flag meanings (apps/btrx:22-60): -f centre frequency (default 2.476e9), -r sample
// rate (required, >= 2e6), -i input file ('-' = stdin), -s input is interleaved int16, -N sample
// limit, -S all-piconet sniffer (default: LAP sniffer), -t SNR squelch (default 10.0),
// -w Wireshark TAP sink.  The little scheduler below stands in for GNU Radio's: history()-1 zeros first,
// work() called with a multiple of output_multiple() new items.  --gpus N (no counterpart in apps/btrx)
// reads the whole capture and time-partitions it over N devices of the node (multi_block::run_partitioned);
// the output is the single-device output.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <gr_bluetooth/multi_LAP.h>
#include <gr_bluetooth/multi_hopper.h>
#include <gr_bluetooth/multi_sniffer.h>

static double eng(const char *s)
{
    char *end = nullptr;
    double v = strtod(s, &end);
    if (end && *end) {
        switch (*end) {
            case 'k': v *= 1e3; break;
            case 'M': v *= 1e6; break;
            case 'G': v *= 1e9; break;
            default: break;
        }
    }
    return v;
}

static void usage()
{
    fprintf(stderr, "usage: btrx_amd -r RATE [-f FREQ] [-i FILE|-] [-s] [-N NSAMPLES] [-S] [-l LAP -p [--aliased]] [-t SNR] [-w] [-c CHUNK_SLOTS] [--gpus N [--all-on-device0]]\n");
}

int main(int argc, char **argv)
{
    double freq = 2.476e9, rate = 0, snr = 10.0, nsamples = -1;
    bool sniff = false, shorts = false, tun = false, hop = false, aliased = false;
    long lap = -1;
    std::string file;
    int chunk_slots = 64, gpus = 1;
    bool all_on_device0 = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *n) { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", n); usage(); exit(1); } return argv[++i]; };
        if (a == "-f" || a == "--freq") freq = eng(need("-f"));
        else if (a == "-r" || a == "--sample-rate") rate = eng(need("-r"));
        else if (a == "-i" || a == "--input-file") file = need("-i");
        else if (a == "-N" || a == "--nsamples") nsamples = eng(need("-N"));
        else if (a == "-t" || a == "--snr") snr = eng(need("-t"));
        else if (a == "-c") chunk_slots = atoi(need("-c"));
        else if (a == "--gpus") gpus = atoi(need("--gpus"));
        else if (a == "--all-on-device0") all_on_device0 = true;
        else if (a == "-S" || a == "--sniff") sniff = true;
        else if (a == "-l" || a == "--lap") lap = strtol(need("-l"), nullptr, 16);     // apps/btrx:42-43
        else if (a == "-p" || a == "--hop") hop = true;                                // apps/btrx:46-47
        else if (a == "--aliased") aliased = true;
        else if (a == "-s" || a == "--input-shorts") shorts = true;
        else if (a == "-w" || a == "--wireshark") tun = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 1; }
    }
    if (rate <= 0) { fprintf(stderr, "Sample rate must be provided\n"); return 1; }             // apps/btrx:70-71
    if (rate < 2e6) { fprintf(stderr, "Sample rate (%d) below minimum (%d)\n", (int)rate, 2000000); return 1; }
    if (file.empty()) { fprintf(stderr, "btrx_amd reads captures only: give -i FILE (or -i -)\n"); return 1; }
    FILE *fp = file == "-" ? stdin : fopen(file.c_str(), "rb");
    if (!fp) { perror(file.c_str()); return 1; }

    std::shared_ptr<gr::bluetooth::multi_block> blk;
    try {
        if (sniff) blk = gr::bluetooth::multi_sniffer::make(rate, freq, snr, tun);
        else if (lap >= 0 && hop) blk = gr::bluetooth::multi_hopper::make(rate, freq, snr, (int)lap, aliased, tun);   // apps/btrx:151-155
        else if (lap >= 0) { fprintf(stderr, "multi_UAP (-l without -p) is not part of this build; use -S or -l LAP -p\n"); return 1; }
        else blk = gr::bluetooth::multi_LAP::make(rate, freq, snr);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 2;
    }
    const size_t H = blk->history(), mult = (size_t)blk->output_multiple();
    std::vector<gr_complex> buf(H - 1, gr_complex(0, 0));          // GNU Radio pre-fills history()-1 zeros
    std::vector<int16_t> sbuf;
    if (gpus > 1) {
        // whole capture in memory, one time range per device
        for (;;) {
            size_t want = (size_t)1 << 22;
            if (nsamples >= 0 && (double)(buf.size() - (H - 1)) + (double)want > nsamples) want = (size_t)(nsamples - (double)(buf.size() - (H - 1)));
            if (want == 0) break;
            size_t old = buf.size(), got;
            buf.resize(old + want);
            if (shorts) {
                sbuf.resize(2 * want);
                got = fread(sbuf.data(), 2 * sizeof(int16_t), want, fp);
                for (size_t i = 0; i < got; i++) buf[old + i] = gr_complex(sbuf[2 * i], sbuf[2 * i + 1]);
            } else got = fread(&buf[old], sizeof(gr_complex), want, fp);
            buf.resize(old + got);
            if (got < want) break;
        }
        if (fp != stdin) fclose(fp);
        try {
            blk->run_partitioned(buf.data(), buf.size() - (H - 1), gpus, all_on_device0);
        } catch (const std::exception &e) {
            fprintf(stderr, "%s\n", e.what());
            return 2;
        }
        return 0;
    }
    const size_t chunk = mult * (size_t)(chunk_slots > 0 ? chunk_slots : 1);
    double remaining = nsamples;
    bool eof = false;
    while (!eof) {
        size_t want = chunk;
        if (remaining >= 0 && (double)want > remaining) want = (size_t)remaining;
        if (want == 0) break;
        size_t old = buf.size(), got;
        buf.resize(old + want);
        if (shorts) {
            sbuf.resize(2 * want);
            got = fread(sbuf.data(), 2 * sizeof(int16_t), want, fp);
            for (size_t i = 0; i < got; i++) buf[old + i] = gr_complex(sbuf[2 * i], sbuf[2 * i + 1]);
        } else {
            got = fread(&buf[old], sizeof(gr_complex), want, fp);
        }
        buf.resize(old + got);
        if (got < want) eof = true;
        if (remaining >= 0) remaining -= (double)got;
        size_t avail = buf.size() - (H - 1);
        size_t n = avail / mult * mult;
        if (n == 0) continue;
        gr_vector_const_void_star in(1, (const void *)buf.data());
        gr_vector_void_star out;
        int consumed = blk->work((int)n, in, out);
        buf.erase(buf.begin(), buf.begin() + consumed);
    }
    if (fp != stdin) fclose(fp);
    return 0;
}

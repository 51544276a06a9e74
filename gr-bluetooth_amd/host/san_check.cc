// san_check.cc -- the host protocol code (classic.cc: classic_packet parsers, FEC, CRC, printing, LE
// packet text, whitening tables) under AddressSanitizer / UndefinedBehaviorSanitizer.  The reference's own
// parsers read past their buffers in places; this build must not.  make -C host san
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "classic.h"

extern "C" int bt_host_crc_check(const uint8_t *symbols, int length, int clock, int type, int uap);
extern "C" int bt_host_decode_print(const uint8_t *symbols, int length, int uap, uint32_t clock, int have27, char *out, int cap);
extern "C" int bt_host_lut(const char *name, uint8_t *out, int cap);

// the GPU hop-table entry points are not part of this check (hopper handlers need a device)
extern "C" {
int btgpu_hopseq_create(uint32_t, int, int, btgpu_hopseq **) { return BTGPU_ENODEVICE; }
void btgpu_hopseq_destroy(btgpu_hopseq *) {}
int btgpu_hopseq_init_candidates(btgpu_hopseq *, int, int, int) { return BTGPU_ENODEVICE; }
int btgpu_hopseq_winnow(btgpu_hopseq *, int, int, int) { return BTGPU_ENODEVICE; }
int btgpu_hopseq_candidates(btgpu_hopseq *, uint32_t *, int) { return BTGPU_ENODEVICE; }
int btgpu_hopseq_lookup(btgpu_hopseq *, const uint32_t *, int, uint8_t *) { return BTGPU_ENODEVICE; }
}

static uint32_t st = 777u;
static uint32_t rnd() { st = st * 1664525u + 1013904223u; return st >> 8; }

int main()
{
    std::vector<char> out(1 << 16);
    uint8_t lut[256];
    if (bt_host_lut("packet::WHITENING_DATA", lut, 256) != 127 || bt_host_lut("classic_packet::INDICES", lut, 256) != 64) return 1;
    for (int trial = 0; trial < 3000; trial++) {
        int len = trial < 400 ? trial : (int)(rnd() % 3200);          // every short length, then random ones
        std::vector<uint8_t> s((size_t)len);                          // exact size: an over-read is a heap overflow
        for (auto &b : s) b = (uint8_t)(rnd() & 1);
        const uint8_t *p = s.empty() ? nullptr : s.data();
        for (int type = 0; type < 16; type++)
            (void)bt_host_crc_check(p, len, (int)(rnd() & 63), type, (int)(rnd() & 255));
        (void)bt_host_decode_print(p, len, (int)(rnd() & 255), rnd() & 0x7ffffff, (int)(rnd() & 1), out.data(), (int)out.size());
        (void)gr::bluetooth::host::le_packet_text(p, len, (int)(rnd() % 79));
    }
    // the sniffer handlers (per-LAP piconet bookkeeping) on a stream of random hits
    {
        gr::bluetooth::host::sniffer_handlers h;
        btgpu_header hd{};
        for (int i = 0; i < 300; i++) {
            int len = 126 + (int)(rnd() % 3000);
            std::vector<uint8_t> s((size_t)len);
            for (auto &b : s) b = (uint8_t)(rnd() & 1);
            btgpu_hit hit{};
            hit.slot = (uint64_t)i * 3; hit.channel = (int)(rnd() % 79); hit.lap = 0x24d952 + (rnd() % 3); hit.nsym = len; hit.snr_db = 20.0;
            for (int c = 0; c < 64; c++) { hd.uap[c] = (uint8_t)rnd(); hd.type[c] = (uint8_t)(rnd() & 15); }
            hd.fec13_ok = (int)(rnd() & 1);
            (void)h.ac(hit, hd, s.data(), len);
        }
    }
    std::printf("host san_check: ok\n");
    return 0;
}

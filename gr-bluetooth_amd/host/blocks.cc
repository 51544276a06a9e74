// blocks.cc -- gr::bluetooth::multi_block / multi_LAP / multi_sniffer over the btgpu C ABI.
// work() keeps the reference's contract (lib/multi_LAP_impl.cc:65-114,
// lib/multi_sniffer_impl.cc:82-166): it reads history()-1 old items + the new ones from
// input_items[0], prints one line per detection in (slot, channel, offset) order and returns
// the number of items consumed -- a whole number of slots (the reference always returns
// exactly one slot; this block consumes every whole slot it was handed).
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <gr_bluetooth/multi_LAP.h>
#include <gr_bluetooth/multi_hopper.h>
#include <gr_bluetooth/multi_sniffer.h>

#include "classic.h"

namespace gr {
namespace bluetooth {

multi_block::multi_block(double sample_rate, double center_freq, double squelch_threshold, int mode, bool hopper)
{
    // multi_sniffer always runs the LE access-address pass after the classic one
    // (lib/multi_sniffer_impl.cc:94,129-149: leok = brok)
    // and hands the sliced symbols of every hit, with the GPU's sweep of its packet header over the
    // 64 clock candidates, to its packet handlers
    // multi_hopper (lib/multi_hopper_impl.cc:52-56) has the sniffer's symbol history but no LE pass
    // multi_LAP prints the LAP of a hit and nothing that depends on the rest of the window
    // (lib/multi_LAP_impl.cc:93-110): no clock-recovery continuation for hit.nsym
    const int flags = mode == BTGPU_MODE_SNIFFER ? ((hopper ? 0 : BTGPU_FLAG_LE) | BTGPU_FLAG_HEADERS | BTGPU_FLAG_EXACT_PAYLOAD) : BTGPU_FLAG_NO_NSYM;
    d_headers = (flags & BTGPU_FLAG_HEADERS) != 0;
    d_sample_rate = sample_rate;
    d_center_freq = center_freq;
    d_target_snr = squelch_threshold;
    d_mode = mode;
    btgpu_config cfg{};
    cfg.sample_rate = sample_rate;
    cfg.center_freq = center_freq;
    cfg.squelch_db = squelch_threshold;
    cfg.mode = mode;
    cfg.device = -1;
    cfg.flags = flags;
    d_cfg = cfg;
    int rc = btgpu_create(&cfg, &d_gpu);
    if (rc != BTGPU_OK)      // no CPU fallback: fail loudly
        throw std::runtime_error(std::string("gr::bluetooth: btgpu_create failed: ") + btgpu_strerror(rc));
    btgpu_get_design(d_gpu, &d_design);
    d_device = btgpu_device(d_gpu);                 // the ordinal cfg.device = -1 resolved to
    // reference: lib/multi_block.cc:116-119
    printf("history set to %d samples: channel=%d, noise=%d\n", d_design.history,
           d_design.ntaps_channel + d_design.decimation * 8, d_design.ntaps_noise);
    set_history((unsigned)d_design.history);
    set_output_multiple(d_design.samples_per_slot);     // the reference's implicit assumption, made explicit
}

multi_block::~multi_block()
{
    if (d_gpu) btgpu_destroy(d_gpu);
}

int multi_block::run_work(int noutput_items, gr_vector_const_void_star &input_items)
{
    const float *in = (const float *)input_items[0];
    size_t consumed = 0;
    int rc = btgpu_work(d_gpu, in, (size_t)(history() - 1) + (size_t)noutput_items, &consumed);
    if (rc == BTGPU_EOVERFLOW)                        // records were dropped: the reference would have printed them
        fprintf(stderr, "Warning: hit buffer overflow, detections of this call were dropped (%s)\n", btgpu_last_error(d_gpu));
    if (rc != BTGPU_OK && rc != BTGPU_EOVERFLOW) {
        // reference convention for fatal errors: fprintf + abort (lib/multi_sniffer_impl.cc:36-40)
        fprintf(stderr, "Error: %s (%s)\n", btgpu_strerror(rc), btgpu_last_error(d_gpu));
        abort();
    }
    drain(d_gpu);
    {
        // the second run's list (hits on rows presence had not made exact) holds 32 768 windows per batch: a window that found it full
        // kept the polyphase path's record -- said once, not silently
        btgpu_timing tm{};
        if (!d_warned_turned_away && btgpu_last_timing(d_gpu, &tm) == BTGPU_OK && tm.verify_turned_away > 0) {
            d_warned_turned_away = true;
            fprintf(stderr, "Warning: %llu windows kept records from rows that are not the reference's arithmetic (the second run's list was full; "
                            "smaller batches: btgpu_config.max_batch_slots)\n", (unsigned long long)tm.verify_turned_away);
        }
    }
    d_cumulative_count += consumed;
    return (int)consumed;
}

void multi_block::drain(btgpu_handle *g)
{
    std::vector<btgpu_hit> buf(256);
    if (d_headers) {
        const int cap = 3125;                         // classic_packet keeps at most MAX_SYMBOLS
        std::vector<uint8_t> syms((size_t)buf.size() * cap);
        std::vector<int> lens(buf.size());
        std::vector<btgpu_header> hdrs(buf.size());
        for (;;) {
            int n = btgpu_poll_headers(g, buf.data(), hdrs.data(), syms.data(), cap, lens.data(), (int)buf.size());
            if (n <= 0) break;
            for (int i = 0; i < n; i++) handle_hit(buf[i], &hdrs[i], syms.data() + (size_t)i * cap, lens[i]);
        }
    } else {
        for (;;) {
            int n = btgpu_poll(g, buf.data(), (int)buf.size());
            if (n <= 0) break;
            for (int i = 0; i < n; i++) handle_hit(buf[i], nullptr, nullptr, 0);
        }
    }
}

long multi_block::run_partitioned(const gr_complex *items, size_t n_new, int ngpus, bool all_on_device0)
{
    const size_t H = (size_t)d_design.history, slot = (size_t)d_design.samples_per_slot, mg = (size_t)d_design.left_margin;
    const uint64_t total = n_new / slot;
    if (ngpus < 1) ngpus = 1;
    if (total == 0) return 0;
    const uint64_t first_abs = d_cumulative_count / slot;              // slot index of the first new slot
    // one handle per range: range 0 on this block's own handle, the others on handles of their own devices
    // the block's own handle sits on d_device (resolved at construction); range r goes to device (d_device + r) % ndev
    const int ndev = btgpu_device_count();
    if (!all_on_device0 && (ndev < ngpus))
        throw std::runtime_error("gr::bluetooth: run_partitioned over " + std::to_string(ngpus) + " devices, " +
                                 std::to_string(ndev < 0 ? 0 : ndev) + " visible");
    std::vector<btgpu_handle *> g((size_t)ngpus, nullptr);
    g[0] = d_gpu;
    for (int r = 1; r < ngpus; r++) {
        btgpu_config cfg = d_cfg;
        cfg.device = all_on_device0 ? d_device : (d_device + r) % ndev;
        int rc = btgpu_create(&cfg, &g[(size_t)r]);
        if (rc != BTGPU_OK) {
            for (int q = 1; q < r; q++) btgpu_destroy(g[(size_t)q]);
            throw std::runtime_error(std::string("gr::bluetooth: btgpu_create on device ") + std::to_string(cfg.device) +
                                     " failed: " + btgpu_strerror(rc));
        }
    }
    std::vector<int> rcs((size_t)ngpus, BTGPU_OK);
    std::vector<std::thread> th;
    const float *base = (const float *)items;                          // items[0] = absolute sample first_abs*slot - (H-1)
    for (int r = 0; r < ngpus; r++) {
        const uint64_t b = total / (uint64_t)ngpus, rem = total % (uint64_t)ngpus;
        const uint64_t first = (uint64_t)r * b + std::min<uint64_t>((uint64_t)r, rem), cnt = b + ((uint64_t)r < rem ? 1 : 0);
        th.emplace_back([=, &rcs, &g]() {
            if (cnt == 0) return;
            // window 0 of the range starts at items[first * slot]; the staged squelch wants mg more samples in front
            const size_t w0 = (size_t)first * slot;
            const size_t have = std::min(mg, w0);                      // the stream start has zeros there (implied)
            int rc = btgpu_process_host(g[(size_t)r], base + 2 * (w0 - have), have + H + (size_t)(cnt - 1) * slot, have,
                                        first_abs + first, cnt);
            if (rc == BTGPU_OK || rc == BTGPU_EOVERFLOW) { int rf = btgpu_flush(g[(size_t)r]); if (rf != BTGPU_OK) rc = rf; }
            rcs[(size_t)r] = rc;
        });
    }
    for (auto &t : th) t.join();
    for (int r = 0; r < ngpus; r++) {
        if (rcs[(size_t)r] == BTGPU_EOVERFLOW)
            fprintf(stderr, "Warning: hit buffer overflow on range %d, detections were dropped\n", r);
        else if (rcs[(size_t)r] != BTGPU_OK) {
            fprintf(stderr, "Error: %s (%s)\n", btgpu_strerror(rcs[(size_t)r]), btgpu_last_error(g[(size_t)r]));
            abort();
        }
    }
    for (int r = 0; r < ngpus; r++) drain(g[(size_t)r]);               // ranges ascend in time: stream order
    for (int r = 1; r < ngpus; r++) btgpu_destroy(g[(size_t)r]);
    d_cumulative_count += total * slot;
    return (long)(total * slot);
}

// ---------------------------------------------------------------- multi_LAP
class multi_LAP_impl : public multi_LAP
{
public:
    multi_LAP_impl(double sample_rate, double center_freq, double squelch_threshold)
        : gr::sync_block("bluetooth multi LAP block", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                         gr::io_signature::make(0, 0, 0)),
          multi_block(sample_rate, center_freq, squelch_threshold, BTGPU_MODE_LAP) {}
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &) override
    {
        return run_work(noutput_items, input_items);
    }
protected:
    void handle_hit(const btgpu_hit &h, const btgpu_header *, const uint8_t *, int) override
    {
        // lib/multi_LAP_impl.cc:97-100
        printf("GOT PACKET: ch=%d, LAP=%06x, err=%u at time slot %d\n", h.channel, h.lap,
               (unsigned)h.ac_errors, (int)h.slot);
    }
};

multi_LAP::sptr multi_LAP::make(double sample_rate, double center_freq, double squelch_threshold)
{
    return gnuradio::get_initial_sptr(new multi_LAP_impl(sample_rate, center_freq, squelch_threshold));
}

// ------------------------------------------------------------ multi_sniffer
class multi_sniffer_impl : public multi_sniffer
{
    bool d_tun;
public:
    multi_sniffer_impl(double sample_rate, double center_freq, double squelch_threshold, bool tun)
        : gr::sync_block("bluetooth multi sniffer block", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                         gr::io_signature::make(0, 0, 0)),
          multi_block(sample_rate, center_freq, squelch_threshold, BTGPU_MODE_SNIFFER), d_tun(tun)
    {
        if (d_tun) {  // lib/multi_sniffer_impl.cc:63-70
            if (d_tap.open("btbb")) d_handlers.set_tap(&d_tap);
            else fprintf(stderr, "warning: was not able to open TUN device, disabling Wireshark interface\n");
        }
    }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &) override
    {
        return run_work(noutput_items, input_items);
    }
protected:
    host::sniffer_handlers d_handlers;
    host::tap_sink d_tap;
    void handle_hit(const btgpu_hit &h, const btgpu_header *hdr, const uint8_t *syms, int nsyms) override
    {
        if (h.kind == BTGPU_KIND_AA) {
            // aa() (lib/multi_sniffer_impl.cc:207-226): "time %6d, snr=%.1f, " + le_packet::print()
            printf("time %6d, snr=%.1f, ", (int)(h.slot & 0x7ffffff), h.snr_db);
            fputs(host::le_packet_text(syms, nsyms, h.channel).c_str(), stdout);
            return;
        }
        // ac() and everything it calls (lib/multi_sniffer_impl.cc:169-365): the "time ..." prefix,
        // then ID, the UAP/CLK1-6 discovery dialogue or the decoded packet
        const std::string text = d_handlers.ac(h, *hdr, syms, nsyms);
        fputs(text.c_str(), stdout);
    }
};

// ------------------------------------------------------------- multi_hopper
class multi_hopper_impl : public multi_hopper
{
    host::hopper_handlers d_handlers;
public:
    multi_hopper_impl(double sample_rate, double center_freq, double squelch_threshold, int LAP, bool aliased, bool tun)
        : gr::sync_block("bluetooth multi hopper block", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                         gr::io_signature::make(0, 0, 0)),
          multi_block(sample_rate, center_freq, squelch_threshold, BTGPU_MODE_SNIFFER, true),
          d_handlers((uint32_t)LAP, aliased, d_design.low_channel, d_design.high_channel)
    {
        if (tun) {    // lib/multi_hopper_impl.cc:59-67
            if (d_tap.open("btbb")) d_handlers.set_tap(&d_tap);
            else fprintf(stderr, "warning: was not able to open TUN device, disabling Wireshark interface\n");
        }
    }
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &) override
    {
        return run_work(noutput_items, input_items);
    }
protected:
    host::tap_sink d_tap;
    void handle_hit(const btgpu_hit &h, const btgpu_header *hdr, const uint8_t *syms, int nsyms) override
    {
        const std::string text = d_handlers.hit(h, *hdr, syms, nsyms);
        if (!text.empty()) fputs(text.c_str(), stdout);
    }
};

multi_hopper::sptr multi_hopper::make(double sample_rate, double center_freq, double squelch_threshold, int LAP,
                                      bool aliased, bool tun)
{
    return gnuradio::get_initial_sptr(new multi_hopper_impl(sample_rate, center_freq, squelch_threshold, LAP, aliased, tun));
}

multi_sniffer::sptr multi_sniffer::make(double sample_rate, double center_freq, double squelch_threshold,
                                        bool tun)
{
    return gnuradio::get_initial_sptr(new multi_sniffer_impl(sample_rate, center_freq, squelch_threshold, tun));
}

}  // namespace bluetooth
}  // namespace gr

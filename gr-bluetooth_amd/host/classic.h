// classic.h -- host half of multi_sniffer's packet handlers:
// the per-LAP piconet bookkeeping and the payload parsers that turn LAP hits into UAP / CLK1-6 and
// printed packets.  The GPU supplies, per hit, the sliced symbols and the header sweep over the 64
// clock candidates (btgpu_poll_headers); everything here is sequential protocol state and stays on
// the host, as in the reference:
//   classic_packet            lib/packet_impl.cc:226-246 (ctor), :367-468 (FEC), :513-548, :597-1202
//   basic_rate_piconet        lib/piconet_impl.cc:433-547 (UAP_from_header, reset), :371-411
//   sniffer_handlers          lib/multi_sniffer_impl.cc:169-365 (ac, id, decode, discover, recall, fhs)
#ifndef GR_BLUETOOTH_AMD_CLASSIC_H
#define GR_BLUETOOTH_AMD_CLASSIC_H

#include <cstdint>
#include <cstdio>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <btgpu.h>

namespace gr {
namespace bluetooth {
namespace host {

// The Wireshark side channel of the reference (lib/tun.cc): Ethernet-framed packets on a TAP device
// "btbb".  open() tries /dev/net/tun like mktun() (:6-79); where that is not possible and the
// environment names a file in BTGPU_TAP_FILE, frames go there instead, each preceded by its length
// (uint32 little endian).  write() = write_interface (:92-123): 14-byte header (dst MAC, src MAC,
// ethertype, big endian) + data.
class tap_sink
{
public:
    ~tap_sink();
    bool open(const char *name);
    bool is_open() const { return d_fd >= 0 || d_file; }
    void write(const uint8_t *data, unsigned len, uint64_t src_addr, uint64_t dst_addr, uint16_t ether_type);
    static std::vector<uint8_t> frame(const uint8_t *data, unsigned len, uint64_t src_addr, uint64_t dst_addr, uint16_t ether_type);
    static constexpr uint16_t ETHER_TYPE = 0xFFF0;       // lib/multi_sniffer_impl.h:52

private:
    int d_fd = -1;
    FILE *d_file = nullptr;
};

class classic_packet
{
public:
    static constexpr int MAX_SYMBOLS = 3125;
    // symbols from the access code on, one per byte; `sweep` = the GPU's try_clock results
    classic_packet(const uint8_t *symbols, int length, uint32_t clkn, int channel, const btgpu_header &sweep);

    uint32_t lap() const { return d_lap; }
    uint32_t clkn() const { return d_clkn; }
    int channel() const { return d_channel; }
    int type() const { return d_type; }
    bool got_payload() const { return d_have_payload; }
    bool header_present() const;                       // lib/packet_impl.cc:1205-1242

    uint8_t try_clock(int clock);                      // :1046-1063, served from the sweep
    int crc_check(int clock);                          // :612-671
    void set_clock(uint32_t clock, bool have27);       // :582-594
    void set_uap(uint8_t uap) { d_uap = uap; }
    void decode(std::string &out);                     // :169-175, :1066-1165
    void print(std::string &out) const;                // :1168-1179
    std::vector<uint8_t> tun_format() const;           // :1181-1210: 6 bytes meta data, 3 bytes header, payload
    void set_nap() { d_have_nap = true; }
    void force_header(int type, uint8_t uap) { d_type = type; d_uap = uap; }   // what a try_clock() left behind (tests)
    int payload_length() const { return d_payload_length; }
    uint8_t uap() const { return d_uap; }
    // FHS payload fields (:1245-1281)
    uint32_t lap_from_fhs() const { return bits(&d_payload[34], 24); }
    uint8_t uap_from_fhs() const { return (uint8_t)bits(&d_payload[64], 8); }
    uint16_t nap_from_fhs() const { return (uint8_t)bits(&d_payload[72], 16); }   // air_to_host8 on 16 bits (Q11)
    uint32_t clock_from_fhs() const { return bits(&d_payload[115], 26); }

    static bool unfec13(const uint8_t *in, uint8_t *out, int length);
    static bool unfec23(const uint8_t *in, int length, std::vector<uint8_t> &out);
    static void unwhiten(const uint8_t *in, uint8_t *out, int clock, int length, int skip);
    static uint16_t crcgen(const uint8_t *payload, int length, int uap);
    static int uap_from_hec(uint16_t data, uint8_t hec);
    static uint32_t bits(const uint8_t *air, int n);

private:
    bool payload_crc() const;
    bool decode_payload_header(const uint8_t *stream, int clock, int header_bytes, int size, bool fec);
    int fhs(int clock);
    int DM(int clock);
    int DH(int clock);
    int EV(int clock, int maxlength);
    int EV4(int clock);
    int HV(int clock);
    bool decode_header(std::string &out);
    void decode_payload();

    std::vector<uint8_t> d_symbols;                    // MAX_SYMBOLS + slack, zero beyond d_length
    int d_length;
    uint32_t d_clkn;
    int d_channel;
    uint32_t d_lap;
    btgpu_header d_sweep;
    int d_type = 0;
    uint8_t d_uap = 0;
    uint32_t d_clock = 0;
    bool d_have_clk6 = false, d_have_clk27 = false, d_have_nap = false;
    bool d_have_payload = false;
    int d_payload_length = 0, d_payload_header_length = 0, d_llid = 0, d_flow = 0;
    uint8_t d_header[18] = {0};
    std::vector<uint8_t> d_payload;                    // one bit per byte
};

// le_packet_impl constructor + print (lib/packet_impl.cc:1529-1664): what aa() prints after its
// "time .., snr=.., " prefix for symbols that start at the LE preamble on classic channel `channel`
std::string le_packet_text(const uint8_t *symbols, int avail, int channel);

class basic_rate_piconet
{
public:
    explicit basic_rate_piconet(uint32_t lap) : d_lap(lap) {}
    ~basic_rate_piconet();
    basic_rate_piconet(const basic_rate_piconet &) = delete;
    basic_rate_piconet &operator=(const basic_rate_piconet &) = delete;
    bool have_uap() const { return d_have_uap; }
    bool have_clk6() const { return d_have_clk6; }
    bool have_clk27() const { return d_have_clk27; }
    bool have_nap() const { return d_have_nap; }
    uint8_t uap() const { return d_uap; }
    uint32_t offset() const { return d_clk_offset; }
    uint16_t nap() const { return d_nap; }
    void set_uap(uint8_t u) { d_uap = u; d_have_uap = true; }
    void set_nap(uint16_t n) { d_nap = n; d_have_nap = true; }
    void set_offset(uint32_t o) { d_clk_offset = o; d_have_clk6 = true; d_have_clk27 = true; }
    bool uap_from_header(classic_packet &pkt, std::string &out);   // lib/piconet_impl.cc:433-517
    void reset(std::string &out);                                  // :526-547
    // hop reversal (lib/piconet_impl.cc:96-129, 279-368): the sequence table, the CLK1-27 candidate
    // list and its winnowing live on the GPU (btgpu_hopseq_*)
    int init_hop_reversal(bool aliased, std::string &out);
    int winnow(std::string &out);
    int winnow(int offset, int channel, std::string &out);
    int hop(uint32_t clock);
    static int aliased_channel(int channel) { return ((channel + 24) % 25) + 26; }
    std::deque<std::shared_ptr<classic_packet>> queue;

private:
    uint32_t d_lap;
    bool d_got_first_packet = false;
    int d_packets_observed = 0, d_total_packets_observed = 0;
    uint32_t d_first_pkt_time = 0;
    uint64_t d_clk6_alive = ~0ull;        // CLK1-6 hypotheses of the first packet that still fit (bit k)
    uint8_t d_clk6_uap[64] = {0};         // the UAP each hypothesis implies
    bool remember_hop(uint32_t clkn, int channel, std::string &out);
    void lock_clk6(int k, uint8_t uap);
    uint32_t d_clk_offset = 0;
    uint8_t d_uap = 0;
    uint16_t d_nap = 0;
    bool d_have_uap = false, d_have_nap = false, d_have_clk6 = false, d_have_clk27 = false;
    int d_pattern_indices[1000] = {0};
    uint8_t d_pattern_channels[1000] = {0};
    int d_winnowed = 0, d_num_candidates = 0;
    bool d_hop_reversal_inited = false, d_aliased = false, d_afh = false, d_looks_like_afh = false;
    btgpu_hopseq *d_hops = nullptr;
};

// gr::bluetooth::multi_hopper's per-slot logic (lib/multi_hopper_impl.cc:76-209) on the hit records
class hopper_handlers
{
public:
    hopper_handlers(uint32_t lap, bool aliased, int low_channel, int high_channel)
        : d_lap(lap & 0xffffff), d_aliased(aliased), d_low(low_channel), d_high(high_channel), d_piconet(lap & 0xffffff) {}
    // records in (slot, channel, offset) order; returns the text the reference prints
    std::string hit(const btgpu_hit &hit, const btgpu_header &sweep, const uint8_t *symbols, int nsymbols);
    const basic_rate_piconet &piconet() const { return d_piconet; }
    void set_tap(tap_sink *t) { d_tap = t; }

private:
    uint32_t d_lap;
    bool d_aliased;
    int d_low, d_high;
    basic_rate_piconet d_piconet;
    uint64_t d_slot = ~0ull;
    int d_last_channel = -1;
    bool d_slot_done = false, d_locked = false;
    tap_sink *d_tap = nullptr;
};

class sniffer_handlers
{
public:
    // one classic hit, in the order work() reports them; returns the text the reference prints
    std::string ac(const btgpu_hit &hit, const btgpu_header &sweep, const uint8_t *symbols, int nsymbols);
    void set_tap(tap_sink *t) { d_tap = t; }

private:
    void id(uint32_t lap, std::string &out);
    tap_sink *d_tap = nullptr;
    void decode(std::shared_ptr<classic_packet> pkt, std::shared_ptr<basic_rate_piconet> pn, bool first_run, std::string &out);
    void discover(std::shared_ptr<classic_packet> pkt, std::shared_ptr<basic_rate_piconet> pn, std::string &out);
    void recall(std::shared_ptr<basic_rate_piconet> pn, std::string &out);
    void fhs(classic_packet &pkt, std::string &out);
    std::map<uint32_t, std::shared_ptr<basic_rate_piconet>> d_piconets;
};

}  // namespace host
}  // namespace bluetooth
}  // namespace gr

// C entry points onto the packet parsers, for differential tests against the oracle
extern "C" {
int bt_host_crc_check(const uint8_t *symbols, int length, int clock, int type, int uap);
int bt_host_decode_print(const uint8_t *symbols, int length, int uap, uint32_t clock, int have27, char *out, int cap);
}
#endif

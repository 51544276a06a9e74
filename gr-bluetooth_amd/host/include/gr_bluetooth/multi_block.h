// gr::bluetooth::multi_block -- MI355X-backed front end base class.
// Same role and constructor signature as the reference's multi_block
// (include/gr_bluetooth/multi_block.h:40-166); the DSP members of the reference
// (filters, DDC maps, M&M state) live on the GPU behind the btgpu C ABI instead.
#pragma once
#include <cstdint>

#include <gr_bluetooth/api.h>
#include <gnuradio/sync_block.h>   // GNU Radio's, or shim/gnuradio/sync_block.h when built without it (-Ishim)

extern "C" {
#include "btgpu.h"
}

namespace gr {
namespace bluetooth {

class GR_BLUETOOTH_API multi_block : virtual public gr::sync_block
{
protected:
    multi_block() {}
    // hopper: multi_hopper's geometry = the sniffer's history without the LE pass
    multi_block(double sample_rate, double center_freq, double squelch_threshold, int mode, bool hopper = false);
    virtual ~multi_block();

    btgpu_handle *d_gpu = nullptr;
    btgpu_design d_design{};
    uint64_t d_cumulative_count = 0;      // total samples consumed (reference: multi_block.h:62)
    double d_sample_rate = 0, d_center_freq = 0, d_target_snr = 0;
    int d_mode = 0;
    bool d_headers = false;               // records come with symbols and the header sweep

    // forwards one scheduler call to btgpu_work(); returns the number of items consumed
    int run_work(int noutput_items, gr_vector_const_void_star &input_items);
    // per-record output, in (slot, channel, offset) order
    // `syms` = the window's sliced symbols from the hit on (multi_sniffer only), `nsyms` of them
    // hdr: the GPU header sweep of a classic hit (multi_sniffer), else nullptr
    virtual void handle_hit(const btgpu_hit &h, const btgpu_header *hdr, const uint8_t *syms, int nsyms) = 0;

public:
    double samples_per_slot() const { return d_design.samples_per_slot; }
    int low_channel() const { return d_design.low_channel; }
    int high_channel() const { return d_design.high_channel; }
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) = 0;
};

}  // namespace bluetooth
}  // namespace gr

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
    bool d_warned_turned_away = false;    // the "second run's list was full" warning has been printed

    // forwards one scheduler call to btgpu_work(); returns the number of items consumed
    int run_work(int noutput_items, gr_vector_const_void_star &input_items);
    // per-record output, in (slot, channel, offset) order
    // `syms` = the window's sliced symbols from the hit on (multi_sniffer only), `nsyms` of them
    // hdr: the GPU header sweep of a classic hit (multi_sniffer), else nullptr
    virtual void handle_hit(const btgpu_hit &h, const btgpu_header *hdr, const uint8_t *syms, int nsyms) = 0;

    void drain(btgpu_handle *g);          // records of `g` -> handle_hit, in order
    btgpu_config d_cfg{};                 // what d_gpu was created with (run_partitioned clones it per device)
    int d_device = 0;                     // HIP ordinal d_gpu lives on

public:
    // Time-partitioned run over `ngpus` devices of this node (no GNU Radio counterpart: a flowgraph hands a
    // block one stream): `items` is what work() would get for the WHOLE capture -- history()-1 old items
    // followed by n_new new ones.  The whole slots are cut into ngpus contiguous ranges; range r goes, with its
    // left halo of history()-1 (+ left_margin) samples, to device r (one btgpu handle and one host thread
    // each, no collective); the ranges' records, each already ordered, are concatenated in range order --
    // i.e. in stream order -- and pass through the same per-record handlers on the calling thread.
    // all_on_device0: every range on device 0 (dry run on a one-GPU box).  Returns the items consumed.
    long run_partitioned(const gr_complex *items, size_t n_new, int ngpus, bool all_on_device0 = false);
    double samples_per_slot() const { return d_design.samples_per_slot; }
    int low_channel() const { return d_design.low_channel; }
    int high_channel() const { return d_design.high_channel; }
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) = 0;
};

}  // namespace bluetooth
}  // namespace gr

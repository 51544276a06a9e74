// gr::bluetooth::multi_hopper -- public factory, same surface as the reference's
// include/gr_bluetooth/multi_hopper.h:42,55: follow one piconet (LAP) through UAP / CLK1-6
// discovery and hop reversal (CLK1-27), then sniff each slot on the predicted channel.
#pragma once
#include <memory>

#include <gr_bluetooth/multi_block.h>

namespace gr {
namespace bluetooth {

class GR_BLUETOOTH_API multi_hopper : virtual public multi_block
{
public:
    typedef std::shared_ptr<multi_hopper> sptr;
    static sptr make(double sample_rate, double center_freq, double squelch_threshold, int LAP, bool aliased, bool tun);
};

}  // namespace bluetooth
}  // namespace gr

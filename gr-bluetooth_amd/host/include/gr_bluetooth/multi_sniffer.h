// gr::bluetooth::multi_sniffer -- public factory, same surface as the reference's
// include/gr_bluetooth/multi_sniffer.h:44,54.
#pragma once
#include <memory>

#include <gr_bluetooth/multi_block.h>

namespace gr {
namespace bluetooth {

class GR_BLUETOOTH_API multi_sniffer : virtual public multi_block
{
public:
    typedef std::shared_ptr<multi_sniffer> sptr;
    static sptr make(double sample_rate, double center_freq, double squelch_threshold, bool tun);
};

}  // namespace bluetooth
}  // namespace gr

// gr::bluetooth::multi_LAP -- public factory, same surface as the reference's
// include/gr_bluetooth/multi_LAP.h:43,53.
#pragma once
#include <memory>

#include <gr_bluetooth/multi_block.h>

namespace gr {
namespace bluetooth {

class GR_BLUETOOTH_API multi_LAP : virtual public multi_block
{
public:
    typedef std::shared_ptr<multi_LAP> sptr;
    static sptr make(double sample_rate, double center_freq, double squelch_threshold);
};

}  // namespace bluetooth
}  // namespace gr

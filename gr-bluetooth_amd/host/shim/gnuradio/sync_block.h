// Minimal stand-in for the parts of GNU Radio's runtime that gr::bluetooth::multi_block touches,
// used ONLY when this tree is built without GNU Radio (it is found through -Ishim).  It exists so
// that OUR block classes (not the reference's sources) compile and can be driven by the
// harness scheduler in btrx_amd.cc, which reproduces the scheduler contract the reference
// relies on: history()-1 zero items before the stream, work() called with at least
// output_multiple() new items, `return value` items consumed.
#pragma once
#include <complex>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {

class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return sptr(new io_signature(min_streams, max_streams, sizeof_stream_item));
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    int sizeof_stream_item(int) const { return d_size; }
private:
    io_signature(int a, int b, int c) : d_min(a), d_max(b), d_size(c) {}
    int d_min, d_max, d_size;
};

class sync_block {
public:
    sync_block() {}
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out)
        : d_name(name), d_in(in), d_out(out) {}
    virtual ~sync_block() {}
    const std::string &name() const { return d_name; }
    unsigned history() const { return d_history; }
    void set_history(unsigned h) { d_history = h; }
    int output_multiple() const { return d_output_multiple; }
    void set_output_multiple(int m) { d_output_multiple = m; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items,
                     gr_vector_void_star &output_items) = 0;
private:
    std::string d_name;
    io_signature::sptr d_in, d_out;
    unsigned d_history = 1;
    int d_output_multiple = 1;
};

}  // namespace gr

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
}

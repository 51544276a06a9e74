"""Digests of the numeric tables the reference HOLDS as literals in lib/packet_impl.cc -- the only
known answers the checkout carries for the integer half of the hot path besides
samples/channel37.dem.  Run in the build container only (reads /root/reference); writes
tests/golden/lut_digests.json = {table: {"len": n, "sha256": hex of the values as bytes}} plus
digests of two forms DERIVED from them (what the kernels and the host actually use):

  derived/classic_first18   for CLK1-6 = 0..63, the first 18 whitening bits
                            WHITENING_DATA[(classic INDICES[clk] + k) % 127], k < 18, bit k of a
                            little-endian uint32 (classic_packet_impl::unwhiten, :513-526)
  derived/le_whiten16       for LE channel index 0..39, WHITENING_DATA[(le INDICES[i] + k) % 127],
                            k < 16, bit k of a little-endian uint16 (le_packet::sniff_aa, :1481-1483)

The tests regenerate every table from its rule in three places (oracle/bt_oracle.c, bt_uap.c; the
product's csrc/design.cc and host/classic.cc) and compare digests: the rules are pinned to the
reference's literals without the literals ever entering the repo.
"""
import hashlib
import json
import os
import re

SRC = "/root/reference/lib/packet_impl.cc"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_tables(text):
    out = {}
    for m in re.finditer(r"const\s+uint8_t\s+(\w+)::(\w+)\[\]\s*=\s*\{([^}]*)\}", text):
        vals = [int(v, 0) for v in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(3))]
        out["%s::%s" % (m.group(1), m.group(2))] = vals
    return out


def digest(vals, width=1):
    b = b"".join(int(v).to_bytes(width, "little") for v in vals)
    return {"len": len(vals), "sha256": hashlib.sha256(b).hexdigest()}


def main():
    t = parse_tables(open(SRC).read())
    want = ["packet::WHITENING_DATA", "classic_packet::INDICES", "classic_packet::PREAMBLE_DISTANCE",
            "classic_packet::BARKER_DISTANCE", "le_packet::PREAMBLE_DISTANCE",
            "le_packet::ACCESS_ADDRESS_DISTANCE_0", "le_packet::ACCESS_ADDRESS_DISTANCE_1",
            "le_packet::ACCESS_ADDRESS_DISTANCE_2", "le_packet::ACCESS_ADDRESS_DISTANCE_3",
            "le_packet::ACCESS_HEADER_DISTANCE_LSB", "le_packet::ACCESS_HEADER_DISTANCE_MSB",
            "le_packet::DATA_HEADER_DISTANCE_LSB", "le_packet::DATA_HEADER_DISTANCE_MSB", "le_packet::INDICES"]
    assert all(k in t for k in want), sorted(t)
    out = {k: digest(t[k]) for k in want}
    w = t["packet::WHITENING_DATA"]
    assert len(w) == 127
    first18 = [sum(w[(i0 + k) % 127] << k for k in range(18)) for i0 in t["classic_packet::INDICES"]]
    le16 = [sum(w[(i0 + k) % 127] << k for k in range(16)) for i0 in t["le_packet::INDICES"]]
    out["derived/classic_first18"] = digest(first18, 4)
    out["derived/le_whiten16"] = digest(le16, 2)
    json.dump(out, open(os.path.join(HERE, "lut_digests.json"), "w"), indent=1, sort_keys=True)
    print({k: v["len"] for k, v in out.items()})


if __name__ == "__main__":
    main()

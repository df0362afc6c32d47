"""Minimal FITS image reader (primary + IMAGE extensions), enough for the 6-HDU files imageplane_disc_image
writes through cfitsio.  TEST INFRASTRUCTURE."""
import numpy as np

_DT = {-64: ">f8", -32: ">f4", 32: ">i4", 16: ">i2", 64: ">i8", 8: "u1"}


def read(path):
    raw = open(path, "rb").read()
    pos, hdus = 0, []
    while pos < len(raw):
        cards = {}
        while True:
            block = raw[pos:pos + 2880]
            pos += 2880
            done = False
            for i in range(0, 2880, 80):
                card = block[i:i + 80].decode("ascii", "replace")
                key = card[:8].strip()
                if key == "END":
                    done = True
                    break
                if card[8:10] == "= ":
                    val = card[10:].split("/")[0].strip()
                    cards[key] = val.strip("'").strip()
            if done:
                break
        naxis = int(cards.get("NAXIS", 0))
        shape = [int(cards[f"NAXIS{i}"]) for i in range(naxis, 0, -1)]
        bitpix = int(cards.get("BITPIX", 8))
        n = int(np.prod(shape)) if shape else 0
        nbytes = n * abs(bitpix) // 8
        data = None
        data_offset = pos
        if n:
            data = np.frombuffer(raw, dtype=_DT[bitpix], count=n, offset=pos).reshape(shape).astype(_DT[bitpix][1:])
        pos += (nbytes + 2879) // 2880 * 2880
        hdus.append({"header": cards, "name": cards.get("EXTNAME", "PRIMARY"), "data": data, "data_offset": data_offset})
    return hdus


def header_cards(path):
    """[[80-character cards of HDU 0], [... of HDU 1], ...] exactly as stored (for byte-level header comparisons)."""
    raw = open(path, "rb").read()
    out = []
    for h in read(path):
        end = h["data_offset"]
        # walk back to the start of this header: it follows the previous HDU's padded data
        start = 0 if not out else out[-1][1]
        cards = [raw[i:i + 80].decode("ascii", "replace") for i in range(start, end, 80)]
        nbytes = 0 if h["data"] is None else h["data"].size * h["data"].dtype.itemsize
        out.append((cards, end + (nbytes + 2879) // 2880 * 2880))
    return [c for c, _ in out]

"""An independent, value-level decoder of the ulc block syntax written from the reference's normative
FormatSpecs.md ALONE (no libulc source, no oracle code): header table (FormatSpecs.md:35-51), code table (:60-71) and
the per-code unpacking rules (:73-141).  It produces the dequantised MDCT coefficients of a block.

The spec leaves the noise generator open ("randomly cycling +/-Level", "[randomly generated +/-1]"), so for
noise coefficients this decoder returns MAGNITUDES and marks them; everything else is exact:
coefficient = Nybble*|Nybble| * Quantizer, Quantizer = 2^-(5+X) or 2^-(5+14+X), run lengths, levels
((X>>1)+1)^2 * Quantizer/4, tail amplitude (Z+1)^2 * Quantizer/16 decaying by 1 - 2^-19*(Y<<4|X)^2 per coefficient
(binary32 arithmetic, one rounding per step as written in the spec's recurrence).
Test infrastructure (tests/test_spec_decoder.py)."""
import numpy as np

# FormatSpecs.md:35-51 — second header nybble -> window sizes as fractions of N (the asterisk does not matter here)
_WINDOWS = {
    0b0010: (2, 2), 0b0011: (2, 2),
    0b0100: (4, 4, 2), 0b0101: (4, 4, 2), 0b0110: (2, 4, 4), 0b0111: (2, 4, 4),
    0b1000: (8, 8, 4, 2), 0b1001: (8, 8, 4, 2), 0b1010: (4, 8, 8, 2), 0b1011: (4, 8, 8, 2),
    0b1100: (2, 8, 8, 4), 0b1101: (2, 8, 8, 4), 0b1110: (2, 4, 8, 8), 0b1111: (2, 4, 8, 8),
}


class SpecError(Exception):
    pass


def _nybbles(block_bytes):
    b = np.asarray(block_bytes, np.uint8)
    out = np.empty(2 * b.size, np.uint8)
    out[0::2] = b & 15          # "low nybble first" is the container convention of the tools; the spec's sequences are in reading order
    out[1::2] = b >> 4
    return out


def decode_block_coefficients(block_bytes, n_chan, block_size):
    """-> (coef [n_chan][block_size] float32, noise_run [n_chan][block_size] int32, nybbles_consumed).
    noise_run is 0 for coded/zero coefficients and the 1-based index of the noise code (run or tail) that produced the
    coefficient otherwise; noise entries of `coef` hold the magnitude."""
    ny = _nybbles(block_bytes)
    pos = 0

    def get():
        nonlocal pos
        if pos >= ny.size:
            raise SpecError("ran off the block")
        v = int(ny[pos]); pos += 1
        return v

    first = get()
    if first & 8:
        second = get()
        if second not in _WINDOWS:
            raise SpecError("header outside the table")        # 0000b/0001b rows do not exist in the table
        sizes = [block_size // d for d in _WINDOWS[second]]
    else:
        sizes = [block_size]
    coef = np.zeros((n_chan, block_size), np.float32)
    noise = np.zeros((n_chan, block_size), np.int32)
    run_id = 0
    f32 = np.float32
    for ch in range(n_chan):
        base = 0
        for S in sizes:
            n = 0
            # "Each channel begins with a nybble specifying the initial quantizer, akin to a silent Fh" (:107)
            pending_escape = True
            quant = None
            done = False
            while n < S and not done:
                v = 0xF if pending_escape else get()
                pending_escape = False
                if v == 0xF:
                    x = get()
                    if x <= 0xD:
                        quant = f32(2.0) ** f32(-(5 + x))
                    elif x == 0xE:
                        y = get()
                        if y <= 0xC:
                            quant = f32(2.0) ** f32(-(5 + 14 + y))
                        elif y == 0xF:
                            done = True                             # Stop: rest zeros (:109-111)
                        else:
                            raise SpecError("unallocated code")
                    else:                                           # Fh,Fh,Z,Y,X: stop with decaying noise (:113-131)
                        if quant is None:
                            raise SpecError("tail noise before any quantizer")
                        z, y, x2 = get(), get(), get()
                        amp = f32((z + 1) ** 2) * quant / f32(16)
                        decay = f32(1.0) - f32(2.0 ** -19) * f32(((y << 4) | x2) ** 2)
                        run_id += 1
                        while n < S:
                            coef[ch, base + n] = amp; noise[ch, base + n] = run_id
                            amp = f32(amp * decay)
                            n += 1
                        done = True
                elif v == 0x0:
                    run = 1 + get()
                    if n + run > S:
                        raise SpecError("zero run past the end")
                    n += run
                elif v == 0x1:
                    y, x = get(), get()
                    run = ((y << 4) | x) + 33
                    if n + run > S:
                        raise SpecError("zero run past the end")
                    n += run
                elif v == 0x8:
                    z, y, x = get(), get(), get()
                    run = ((z << 5) | (y << 1) | (x & 1)) + 16
                    level = f32(((x >> 1) + 1) ** 2) * quant / f32(4)
                    if n + run > S:
                        raise SpecError("noise run past the end")
                    run_id += 1
                    coef[ch, base + n: base + n + run] = level
                    noise[ch, base + n: base + n + run] = run_id
                    n += run
                else:
                    s = v - 16 if v >= 8 else v                       # -7..-2, +2..+7
                    coef[ch, base + n] = f32(s * abs(s)) * quant
                    n += 1
            base += S
    return coef, noise, pos

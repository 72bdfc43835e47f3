"""eqs_to_sage_mat_helper (gf2bv/_internal.c:678-765): the coefficient matrix as a two-colour PNG for Sage's
unpickle_matrix_mod2_dense_v2 (entry = 1 - palette index, pixel (x, y) = (column, row)) + the affine bits.
No Sage and no libgd in this image: the PNG is decoded here by hand -- signature, chunk CRCs, IHDR, PLTE, zlib, filters --
and compared with the equation ints bit for bit.  CPU-only (host code)."""
import random
import struct
import zlib

import pytest

from gf2bv_amd import LinearSystem, _internal, eqs_to_sage_mat_helper


def decode_png(buf: bytes):
    """-> (width, height, bit depth, colour type, palette [(r, g, b)], rows of palette indices)"""
    assert buf[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(buf):
        n, typ = struct.unpack(">I4s", buf[at:at + 8])
        data = buf[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", buf[at + 8 + n:at + 12 + n])
        assert zlib.crc32(typ + data) == crc, typ
        chunks.append((typ, data))
        at += 12 + n
    assert at == len(buf) and chunks[0][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, depth, ctype, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (comp, flt, lace) == (0, 0, 0)
    plte = [c for t, c in chunks if t == b"PLTE"]
    assert len(plte) == 1 and [t for t, _ in chunks].index(b"PLTE") < [t for t, _ in chunks].index(b"IDAT")
    palette = [tuple(plte[0][i:i + 3]) for i in range(0, len(plte[0]), 3)]
    raw = zlib.decompress(b"".join(c for t, c in chunks if t == b"IDAT"))
    rb = (w * depth + 7) // 8
    assert len(raw) == h * (rb + 1)
    rows = []
    for y in range(h):
        line = raw[y * (rb + 1):(y + 1) * (rb + 1)]
        assert line[0] == 0                                 # filter type None (the only one the writer uses)
        px = []
        for x in range(w):
            bit = x * depth
            px.append((line[1 + bit // 8] >> (8 - depth - bit % 8)) & ((1 << depth) - 1))
        rows.append(px)
    return w, h, depth, ctype, palette, rows


@pytest.mark.parametrize("rows,cols,seed", [(1, 1, 0), (3, 7, 1), (5, 8, 2), (9, 9, 3), (40, 64, 4), (33, 65, 5), (20, 191, 6),
                                            (300, 1000, 7)])
def test_png_decodes_back_to_the_matrix(rows, cols, seed):
    rng = random.Random(seed)
    eqs = [rng.getrandbits(cols + 1) for _ in range(rows)]
    eqs[0] |= 1 << cols                                      # the last column is there
    if rows > 2:
        eqs[1] = 1                                           # affine bit only
        eqs[2] = -(eqs[2] | 2)                               # the sign is ignored (_internal.c:41-59 reads |v|'s digits)
    buf, affine = eqs_to_sage_mat_helper(eqs, cols)
    assert isinstance(buf, bytes) and isinstance(affine, list)
    assert affine == [bool(abs(e) & 1) for e in eqs] and all(type(a) is bool for a in affine)
    w, h, depth, ctype, palette, px = decode_png(buf)
    assert (w, h) == (cols, rows)
    assert ctype == 3 and palette[:2] == [(0, 0, 0), (255, 255, 255)]     # index 0 = black = a set bit (Sage: 1 - index)
    for i, e in enumerate(eqs):
        coeff = abs(e) >> 1
        assert [1 - v for v in px[i]] == [(coeff >> c) & 1 for c in range(cols)], i


def test_bits_above_cols_are_ignored_and_big_rows_cross_stored_block_and_chunk_limits():
    cols = 70000                                             # 8751 bytes per scanline: several 64 KiB stored blocks, > 1 MiB of IDAT
    rng = random.Random(9)
    eqs = [rng.getrandbits(cols + 40) for _ in range(130)]
    buf, affine = eqs_to_sage_mat_helper(eqs, cols)
    w, h, depth, ctype, palette, px = decode_png(buf)
    assert (w, h, depth, ctype) == (cols, 130, 1, 3)
    assert buf.count(b"IDAT") >= 2
    for i in (0, 64, 129):
        coeff = eqs[i] >> 1
        assert [1 - v for v in px[i]] == [(coeff >> c) & 1 for c in range(cols)]
    assert affine == [bool(e & 1) for e in eqs]


def test_argument_errors_follow_the_reference():
    with pytest.raises(TypeError, match="requires 2 arguments"):
        eqs_to_sage_mat_helper([1])
    with pytest.raises(TypeError, match="must be a list"):
        eqs_to_sage_mat_helper((1, 2), 3)
    with pytest.raises(ValueError, match="must be positive"):
        eqs_to_sage_mat_helper([1], 0)
    with pytest.raises(TypeError, match="must be integers"):
        eqs_to_sage_mat_helper([1, "x"], 3)
    buf, affine = eqs_to_sage_mat_helper([], 5)             # no rows: an empty image is still a PNG
    assert decode_png(buf)[:2] == (5, 0) and affine == []
    buf, affine = eqs_to_sage_mat_helper([0, 6], 2)          # the int 0: the reference leaves a NULL slot, here False
    assert affine == [False, False] and [1 - v for v in decode_png(buf)[5][1]] == [1, 1]


def test_linear_system_front_end_uses_it_where_sage_is_present():
    lin = LinearSystem([4, 4])
    a, b = lin.gens()
    zeros = [a ^ b ^ 5, (a >> 1) ^ 3]
    eqs = lin.get_eqs(zeros)
    buf, affine = _internal.eqs_to_sage_mat_helper(eqs, 8)
    px = decode_png(buf)[5]
    assert [[1 - v for v in r] for r in px] == [[(e >> (c + 1)) & 1 for c in range(8)] for e in eqs]
    try:
        import sage.all  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            lin.get_sage_mat(zeros)                          # (the reference imports Sage at call time too, :198-201)
    else:
        A, bb = lin.get_sage_mat(zeros)
        A2, bb2 = lin.get_sage_mat_slow(zeros)
        assert A == A2 and bb == bb2

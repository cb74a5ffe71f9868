"""Minimal PNG reader for heightmap images (reference mjcf.py:490-495 reads them with imageio, which is not a dependency
here): 8- or 16-bit greyscale / RGB / with alpha, non-interlaced.  Returns an integer array [rows, cols(, channels)]."""
import struct
import zlib

import numpy as np


def imread(path):
    raw = open(path, 'rb').read()
    assert raw[:8] == b'\x89PNG\r\n\x1a\n', f'{path} is not a PNG file'
    pos, idat, hdr = 8, b'', None
    while pos < len(raw):
        n, kind = struct.unpack('>I4s', raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif kind == b'IDAT':
            idat += body
        elif kind == b'IEND':
            break
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth in (8, 16) and ctype in (0, 2, 4, 6) and not interlace, 'unsupported PNG variant (need 8/16-bit, no palette, no interlace)'
    nch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    bpp = nch*depth//8
    stride = w*bpp
    data = zlib.decompress(idat)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for r in range(h):
        f = data[r*(stride + 1)]
        line = np.frombuffer(data, np.uint8, stride, r*(stride + 1) + 1).astype(np.int32)
        cur = np.zeros(stride, np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:                      # filters that look left: byte by byte
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b)//2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2*c)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[r] = cur
        prev = cur
    img = out.view('>u2').astype(np.uint16) if depth == 16 else out
    img = img.reshape(h, w, nch)
    return img[:, :, 0] if nch == 1 else img


def imwrite_gray(path, img):
    """Write a greyscale PNG (tests and examples: the reader's counterpart)."""
    img = np.asarray(img)
    depth = 16 if img.dtype == np.uint16 else 8
    rows = img.astype('>u2' if depth == 16 else np.uint8)
    body = b''.join(b'\x00' + rows[r].tobytes() for r in range(rows.shape[0]))

    def chunk(kind, data):
        c = struct.pack('>I', len(data)) + kind + data
        return c + struct.pack('>I', zlib.crc32(kind + data) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', img.shape[1], img.shape[0], depth, 0, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(body)) + chunk(b'IEND', b''))

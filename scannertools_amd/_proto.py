"""Tiny proto3 wire-format writer for the ops' argument messages
(/root/reference/scannertools/scannertools_cpp/imgproc/scannertools_imgproc.proto); the C++ side
reads them with scanner_kernels/proto_lite.h.  Default-valued fields are omitted, as proto3 does."""
import struct


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def encode(fields):
    """fields: iterable of (number, kind, value), kind in int32 / bool / float / string."""
    out = bytearray()
    for number, kind, value in fields:
        if kind in ("int32", "bool"):
            if int(value) == 0:
                continue
            out += _varint(number << 3 | 0) + _varint(int(value))
        elif kind == "float":
            if float(value) == 0.0:
                continue
            out += _varint(number << 3 | 5) + struct.pack("<f", float(value))
        elif kind == "string":
            data = value.encode() if isinstance(value, str) else bytes(value)
            if not data:
                continue
            out += _varint(number << 3 | 2) + _varint(len(data)) + data
        else:
            raise ValueError("unsupported field kind %r" % kind)
    return bytes(out)

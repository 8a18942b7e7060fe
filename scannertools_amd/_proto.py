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


def fields(buf):
    """Iterate a serialized message: yields (number, wire_type, value); value is an int for varint / fixed
    fields and a memoryview for length-delimited ones.  Raises ValueError on a truncated message."""
    mv = memoryview(buf)
    i, n = 0, len(mv)
    while i < n:
        key, shift = 0, 0
        while True:
            if i >= n:
                raise ValueError("truncated varint")
            b = mv[i]
            i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        number, wt = key >> 3, key & 7
        if wt == 0:
            v, shift = 0, 0
            while True:
                if i >= n:
                    raise ValueError("truncated varint")
                b = mv[i]
                i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield number, wt, v
        elif wt == 1:
            if i + 8 > n:
                raise ValueError("truncated fixed64")
            yield number, wt, int.from_bytes(mv[i:i + 8], "little")
            i += 8
        elif wt == 5:
            if i + 4 > n:
                raise ValueError("truncated fixed32")
            yield number, wt, int.from_bytes(mv[i:i + 4], "little")
            i += 4
        elif wt == 2:
            ln, shift = 0, 0
            while True:
                if i >= n:
                    raise ValueError("truncated length")
                b = mv[i]
                i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            if i + ln > n:
                raise ValueError("truncated field %d" % number)
            yield number, wt, mv[i:i + ln]
            i += ln
        else:
            raise ValueError("unsupported wire type %d" % wt)


def message(number, payload):
    """A length-delimited field (sub-message, bytes, packed repeated) with the given serialized payload."""
    return _varint(number << 3 | 2) + _varint(len(payload)) + bytes(payload)

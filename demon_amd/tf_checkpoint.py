"""Pure-Python reader (and minimal writer) of TensorFlow "TensorBundle" checkpoints -- the format of the
reference's weights `weights/demon_original.{index,data-00000-of-00001}` that `tf.train.Saver().restore` loads in
examples/example.py:82-83.  No TensorFlow needed.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table):
  <prefix>.index   an immutable sorted string table (LevelDB table format):
       [data block]* [metaindex block] [index block] [footer: metaindex handle, index handle, pad to 40 B, magic]
       block   = entries (varint32 shared, varint32 non_shared, varint32 value_len, key suffix, value)*,
                 uint32 restarts[], uint32 num_restarts;  followed by 1 byte compression type + 4 byte masked crc32c
       key ""  -> BundleHeaderProto,  key <tensor name> -> BundleEntryProto
                 (1 dtype, 2 shape{2 dim{1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c)
  <prefix>.data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset + size)

The writer exists so that tests can round-trip the reader and so that synthetic weights can be shipped in the
reference's own format; it writes uncompressed blocks like TensorFlow's BundleWriter does.
"""
import os
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DT_FLOAT = 1
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}


# ---- varints / minimal protobuf ----------------------------------------------------------------------------
def _read_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _write_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field number, wire type, value) ; value is int (varint / fixed) or bytes (length delimited)"""
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((field, wt, v))
    return out


def _parse_entry(value):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0}
    for field, wt, v in _parse_proto(value):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:  # Dim
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v   # masked crc32c of the tensor bytes (fixed32); TensorFlow's BundleReader verifies it
        elif field == 7:
            raise ValueError("sliced (partitioned) variables are not supported")
    return e


# ---- table reader ------------------------------------------------------------------------------------------------
def _read_block(data, offset, size):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed table blocks (type %d) are not supported; TensorFlow writes bundle indices uncompressed" % ctype)
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(prefix):
    """-> dict tensor name -> entry dict (dtype, shape, shard_id, offset, size), plus the header under ''"""
    with open(prefix + ".index", "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != _MAGIC:
        raise ValueError("%s.index is not a TensorBundle index (bad magic)" % prefix)
    footer = data[len(data) - 48:]
    pos = 0
    _, pos = _read_varint(footer, pos)   # metaindex offset
    _, pos = _read_varint(footer, pos)   # metaindex size
    idx_off, pos = _read_varint(footer, pos)
    idx_size, pos = _read_varint(footer, pos)
    entries = {}
    for _, handle in _read_block(data, idx_off, idx_size):
        boff, p = _read_varint(handle, 0)
        bsize, p = _read_varint(handle, p)
        for key, value in _read_block(data, boff, bsize):
            entries[key.decode()] = value
    header = entries.pop("", None)
    out = {name: _parse_entry(v) for name, v in entries.items()}
    num_shards = 1
    if header is not None:
        for field, _, v in _parse_proto(header):
            if field == 1:
                num_shards = v
            if field == 2 and v != 0:
                raise ValueError("big-endian bundles are not supported")
    return out, num_shards


def load_tf_checkpoint(prefix, names=None, verify_crc=True):
    """Reads the float variables of a TF checkpoint: dict name -> numpy array (TF layout).  `names` restricts /
    validates the set (KeyError when one is missing); optimizer slots etc. are otherwise returned too."""
    index, num_shards = read_index(prefix)
    wanted = list(index) if names is None else list(names)
    out = {}
    files = {}
    try:
        for name in wanted:
            if name not in index:
                raise KeyError("variable %s not found in checkpoint %s" % (name, prefix))
            e = index[name]
            if e["dtype"] not in _DTYPES:
                if names is None:
                    continue
                raise ValueError("variable %s has unsupported dtype %d" % (name, e["dtype"]))
            dt = np.dtype(_DTYPES[e["dtype"]])
            shard = e["shard_id"]
            if shard not in files:
                files[shard] = open("%s.data-%05d-of-%05d" % (prefix, shard, num_shards), "rb")
            f = files[shard]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            count = int(np.prod(e["shape"])) if e["shape"] else 1
            if len(raw) != e["size"] or e["size"] != count * dt.itemsize:
                raise ValueError("variable %s: size mismatch in data file" % name)
            if verify_crc and "crc32c" in e and _mask_crc(crc32c(raw)) != e["crc32c"]:
                raise ValueError("variable %s: checksum does not match (data file corrupt)" % name)
            out[name] = np.frombuffer(raw, dtype=dt.newbyteorder("<")).reshape(e["shape"]).astype(dt)
    finally:
        for f in files.values():
            f.close()
    return out


# ---- minimal writer (tests, shipping synthetic weights in the reference's format) -------------------------------------
def _mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


_CRC_TABLE = None
_CRC_CHUNK = 4096
_CRC_ZERO_TABLES = None   # the operator "append _CRC_CHUNK zero bytes" on the raw crc register, as 4 x 256 lookup tables


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl[i] = c
        _CRC_TABLE = tbl
    return _CRC_TABLE


def _crc_zero_tables():
    """register' = M * register for one chunk of zero bytes; M is linear over GF(2), so it is tabulated per register byte"""
    global _CRC_ZERO_TABLES
    if _CRC_ZERO_TABLES is None:
        tbl = _crc_table()
        basis = (np.uint32(1) << np.arange(32, dtype=np.uint32)).astype(np.uint32)
        for _ in range(_CRC_CHUNK):
            basis = tbl[basis & 0xFF] ^ (basis >> np.uint32(8))
        tables = np.zeros((4, 256), np.uint32)
        for byte in range(4):
            for v in range(256):
                acc = np.uint32(0)
                for bit in range(8):
                    if v >> bit & 1:
                        acc ^= basis[8 * byte + bit]
                tables[byte, v] = acc
        _CRC_ZERO_TABLES = tables
    return _CRC_ZERO_TABLES


def _crc_register(data, reg):
    """raw crc32c register after feeding `data` (uint8 array), scalar table loop -- for short tails only"""
    tbl = _crc_table()
    reg = int(reg)
    for b in data.tolist():
        reg = int(tbl[(reg ^ b) & 0xFF]) ^ (reg >> 8)
    return reg


def crc32c(data):
    """CRC-32C (Castagnoli) of a bytes-like object.  Tensors are hundreds of MB, so the byte recurrence runs vectorised over
    4 KB chunks (numpy: all chunks advance one byte per step), and the chunk registers are chained with the linear
    "append 4096 zero bytes" operator: R(A || B, x) = M_|B| R(A, x) xor R(B, 0)."""
    buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8)
    nfull = len(buf) // _CRC_CHUNK
    reg = 0xFFFFFFFF
    if nfull:
        tbl = _crc_table()
        chunks = buf[:nfull * _CRC_CHUNK].reshape(nfull, _CRC_CHUNK)
        regs = np.zeros(nfull, np.uint32)
        for j in range(_CRC_CHUNK):
            regs = tbl[(regs ^ chunks[:, j]) & 0xFF] ^ (regs >> np.uint32(8))
        z = _crc_zero_tables()
        for c in regs.tolist():
            reg = int(z[0, reg & 0xFF]) ^ int(z[1, (reg >> 8) & 0xFF]) ^ int(z[2, (reg >> 16) & 0xFF]) ^ int(z[3, reg >> 24]) ^ c
    reg = _crc_register(buf[nfull * _CRC_CHUNK:], reg)
    return reg ^ 0xFFFFFFFF


def _crc32c(data):
    return crc32c(data)


def _build_block(items):
    """one restart point per entry (no prefix sharing) -- valid and simple"""
    body, restarts = bytearray(), []
    for key, value in items:
        restarts.append(len(body))
        body += _write_varint(0) + _write_varint(len(key)) + _write_varint(len(value)) + key + value
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _entry_proto(dtype, shape, offset, size, crc):
    """BundleEntryProto: dtype (1), shape (2), offset (4), size (5), crc32c (6, fixed32, masked) -- TensorFlow's BundleReader
    rejects entries whose checksum is missing or wrong ("Checksum does not match")"""
    dims = b"".join(b"\x12" + _write_varint(len(d)) + d for d in (b"\x08" + _write_varint(s) for s in shape))
    return (b"\x08" + _write_varint(dtype) + b"\x12" + _write_varint(len(dims)) + dims +
            b"\x20" + _write_varint(offset) + b"\x28" + _write_varint(size) + b"\x35" + struct.pack("<I", _mask_crc(crc)))


def save_tf_checkpoint(prefix, tensors, block_size=4096):
    """Writes dict name -> float32 array as a single-shard TensorBundle."""
    names = sorted(tensors)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01")]  # BundleHeaderProto{num_shards: 1, version{producer: 1}}
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in names:
            raw = np.asarray(tensors[name], dtype="<f4").tobytes(order="C")
            f.write(raw)
            a = np.asarray(tensors[name])
            items.append((name.encode(), _entry_proto(_DT_FLOAT, a.shape, offset, len(raw), crc32c(raw))))
            offset += len(raw)
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)  # no compression
        out.extend(struct.pack("<I", _mask_crc(_crc32c(block + b"\x00"))))
        return off, len(block)

    index_items, cur, cur_bytes = [], [], 0
    for key, value in items:
        cur.append((key, value))
        cur_bytes += len(key) + len(value) + 3
        if cur_bytes >= block_size:
            off, size = emit(_build_block(cur))
            index_items.append((cur[-1][0], _write_varint(off) + _write_varint(size)))
            cur, cur_bytes = [], 0
    if cur:
        off, size = emit(_build_block(cur))
        index_items.append((cur[-1][0], _write_varint(off) + _write_varint(size)))
    meta_off, meta_size = emit(_build_block([]))
    idx_off, idx_size = emit(_build_block(index_items))
    footer = _write_varint(meta_off) + _write_varint(meta_size) + _write_varint(idx_off) + _write_varint(idx_size)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))

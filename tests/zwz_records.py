""".zwz record (de)serialiser for tests and fixture minting (compression.cpp:73-104).
Test infrastructure only: derives damaged / reordered shards from a good one, byte for byte reproducibly."""
import struct


def parse(blob):
    """-> list of (path, seq, last, payload, md5 or None); stops at the first incomplete record."""
    out, p = [], 0
    while p + 8 <= len(blob):
        total, path_len = struct.unpack_from("<ii", blob, p)
        q = p + 8
        if path_len < 0 or q + path_len + 5 > len(blob):
            break
        path = blob[q:q + path_len]; q += path_len
        seq, last = struct.unpack_from("<iB", blob, q); q += 5
        plen = total - (4 + path_len + 4 + 1)
        if plen < 0 or q + plen > len(blob):
            break
        payload = blob[q:q + plen]; q += plen
        md5 = None
        if last:
            if q + 32 > len(blob):
                break
            md5 = blob[q:q + 32]; q += 32
        out.append((path, seq, last, payload, md5))
        p = q
    return out


def serialise(records):
    out = bytearray()
    for path, seq, last, payload, md5 in records:
        out += struct.pack("<ii", 4 + len(path) + 4 + 1 + len(payload), len(path)) + path
        out += struct.pack("<iB", seq, last) + payload
        if last:
            out += md5
    return bytes(out)


def record_spans(blob):
    """-> list of (start, payload_start, md5_start or None, end) byte offsets of every complete record."""
    out, p = [], 0
    while p + 8 <= len(blob):
        total, path_len = struct.unpack_from("<ii", blob, p)
        q = p + 8 + path_len
        if q + 5 > len(blob):
            break
        last = blob[q + 4]
        q += 5
        ps = q
        q += total - (4 + path_len + 4 + 1)
        ms = q if last else None
        if last:
            q += 32
        if q > len(blob):
            break
        out.append((p, ps, ms, q))
        p = q
    return out


def edge_shards(good):
    """Damaged / reordered variants of a good shard, as {name: bytes}.  What each one exercises:
      permuted      records of multi-chunk files out of order (decompression.cpp:119-153: expected id + pending heap),
                    incl. a last chunk that arrives early (its file is then never finalised: no MD5 check)
      cut_boundary  the shard ends exactly after a record
      cut_md5       the shard ends inside an MD5 trailer
      cut_payload   the shard ends inside a payload
      cut_header    the shard ends inside a record header
      empty         zero bytes
    """
    recs = parse(good)
    by_path = {}
    for r in recs:
        by_path.setdefault(r[0], []).append(r)
    perm = []
    done = set()
    for r in recs:
        path = r[0]
        if path in done:
            continue
        done.add(path)
        group = by_path[path]
        if len(group) == 2:
            group = [group[1], group[0]]            # last chunk first: never finalised by the reference
        elif len(group) >= 3:
            group = [group[1], group[0]] + group[2:]   # a middle swap: drained from the pending heap, finalised normally
        perm += group
    spans = record_spans(good)
    k = len(spans) // 2
    md5_spans = [s for s in spans if s[2] is not None]
    big = max(spans, key=lambda s: s[3] - s[1])
    return {
        "permuted": serialise(perm),
        "cut_boundary": good[:spans[k][3]],
        "cut_md5": good[:md5_spans[len(md5_spans) // 2][2] + 11],
        "cut_payload": good[:big[1] + (big[3] - big[1]) // 3],
        "cut_header": good[:spans[k][0] + 6],
        "empty": b"",
    }

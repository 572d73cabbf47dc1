"""Spec-level pure-Python big-int restatement of the signature-verification path.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and only as
the checker.  It exists to pin the C oracle (oracle/secp256k1_oracle.c) and the HIP
path against an implementation that is short enough to audit against the specs.

What it restates (reference = /root/reference, Core Lightning v26.06.6):
  * check_signed_hash()            bitcoin/signature.c:174-192  -> ecdsa_verify()
  * check_signed_hash_nodeid()     common/node_id.c:72-80       -> ecdsa_verify_der33()
  * check_schnorr_sig()            bitcoin/signature.c:408-430  -> schnorr_verify()
  * pubkey_from_der()/node_id      bitcoin/pubkey.c:14-24, common/node_id.c:21-27 -> pubkey_parse()
  * fromwire_secp256k1_ecdsa_signature()  wire/fromwire.c:188-199 -> sig_parse_compact()
  * signature_from_der()           bitcoin/signature.c:310-323  -> sig_parse_der()
  * sha256_double()                bitcoin/shadouble.c:7-11     -> sha256d()
  * sigcheck_channel_announcement/_channel_update/_node_announcement
                                   gossipd/sigcheck.c:9-164     -> sigcheck_*()
  * bitcoin_tx_hash_for_sig()      bitcoin/signature.c:120-151 (libwally BIP143) -> bip143_sighash()

The arithmetic itself lives in libsecp256k1-zkp (libwally-core 1.4.0's nested
submodule), which is ABSENT from /root/reference (empty submodule).  Its behaviour
is restated here from the published algorithms: SEC1/SEC2 (curve + ECDSA), BIP-62
low-S rule as enforced by secp256k1_ecdsa_verify, BIP-340 (Schnorr), BIP-143.
"""
import hashlib

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)
HALF_N = N >> 1


# ---------------------------------------------------------------- group law (affine, None = infinity)
def on_curve(pt):
    x, y = pt
    return (y * y - x * x * x - 7) % P == 0


def padd(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def pneg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def pmul(k, pt):
    k %= N
    acc = None
    while k:
        if k & 1:
            acc = padd(acc, pt)
        pt = padd(pt, pt)
        k >>= 1
    return acc


def lift_x(x):
    """BIP-340 lift_x: the point with this x and EVEN y, or None."""
    if x >= P:
        return None
    c = (pow(x, 3, P) + 7) % P
    y = pow(c, (P + 1) // 4, P)
    if y * y % P != c:
        return None
    return (x, y if y % 2 == 0 else P - y)


# ---------------------------------------------------------------- parsing
def pubkey_parse(b):
    """secp256k1_ec_pubkey_parse: 33 B (02/03) or 65 B (04, hybrid 06/07).  None on failure."""
    if len(b) == 33 and b[0] in (2, 3):
        x = int.from_bytes(b[1:], "big")
        pt = lift_x(x)
        if pt is None:
            return None
        if (pt[1] & 1) != (b[0] & 1):
            pt = (pt[0], P - pt[1])
        return pt
    if len(b) == 65 and b[0] in (4, 6, 7):
        x = int.from_bytes(b[1:33], "big")
        y = int.from_bytes(b[33:], "big")
        if x >= P or y >= P:
            return None
        if b[0] in (6, 7) and (y & 1) != (b[0] & 1):
            return None
        if not on_curve((x, y)):
            return None
        return (x, y)
    return None


def sig_parse_compact(sig64):
    """secp256k1_ecdsa_signature_parse_compact: None iff r >= n or s >= n (wire/fromwire.c:196)."""
    r = int.from_bytes(sig64[:32], "big")
    s = int.from_bytes(sig64[32:], "big")
    if r >= N or s >= N:
        return None
    return (r, s)


def _der_read_len(b, pos, end):
    if pos >= end:
        return None
    b1 = b[pos]
    pos += 1
    if b1 == 0xFF:
        return None
    if b1 & 0x80 == 0:
        return b1, pos
    if b1 == 0x80:
        return None
    lenleft = b1 & 0x7F
    if lenleft > end - pos:
        return None
    if b[pos] == 0:
        return None  # not the shortest encoding
    if lenleft > 8:
        return None
    ret = 0
    while lenleft > 0:
        ret = (ret << 8) | b[pos]
        pos += 1
        lenleft -= 1
    if ret > end - pos:
        return None
    if ret < 128:
        return None  # not the shortest encoding
    return ret, pos


def _der_parse_integer(b, pos, end):
    """-> (value, newpos) or None.  Out-of-range (negative, >32 bytes, >= n) parses as 0."""
    if pos == end or b[pos] != 0x02:
        return None
    pos += 1
    rl = _der_read_len(b, pos, end)
    if rl is None:
        return None
    rlen, pos = rl
    if rlen == 0 or rlen > end - pos:
        return None
    if b[pos] == 0x00 and rlen > 1 and (b[pos + 1] & 0x80) == 0:
        return None  # excessive 0x00 padding
    if b[pos] == 0xFF and rlen > 1 and (b[pos + 1] & 0x80) == 0x80:
        return None  # excessive 0xFF padding
    overflow = bool(b[pos] & 0x80)  # negative
    start, ln = pos, rlen
    if ln > 0 and b[start] == 0:
        ln -= 1
        start += 1
    if ln > 32:
        overflow = True
    val = 0
    if not overflow:
        val = int.from_bytes(b[start:start + ln], "big")
        if val >= N:
            overflow = True
    if overflow:
        val = 0
    return val, pos + rlen


def sig_parse_der(der):
    """secp256k1_ecdsa_signature_parse_der (strict DER).  -> (r, s) or None.
    As upstream: an out-of-range integer still *parses* (as 0) and is rejected at verify."""
    b = bytes(der)
    end = len(b)
    pos = 0
    if pos == end or b[pos] != 0x30:
        return None
    pos += 1
    rl = _der_read_len(b, pos, end)
    if rl is None:
        return None
    rlen, pos = rl
    if rlen != end - pos:
        return None
    ri = _der_parse_integer(b, pos, end)
    if ri is None:
        return None
    r, pos = ri
    si = _der_parse_integer(b, pos, end)
    if si is None:
        return None
    s, pos = si
    if pos != end:
        return None
    return (r, s)


SIGHASH_ALL = 1
SIGHASH_SINGLE_ANYONECANPAY = 0x83


def signature_from_der(der):
    """bitcoin/signature.c:310-323: DER body + 1 trailing sighash byte.  -> ((r,s), sighash_type) or None."""
    if len(der) < 1:
        return None
    rs = sig_parse_der(der[:-1])
    if rs is None:
        return None
    if der[-1] not in (SIGHASH_ALL, SIGHASH_SINGLE_ANYONECANPAY):
        return None
    return rs, der[-1]


# ---------------------------------------------------------------- hashes
def sha256(b):
    return hashlib.sha256(b).digest()


def sha256d(b):
    return sha256(sha256(b))


def tagged_hash(tag, msg):
    t = sha256(tag.encode())
    return sha256(t + t + msg)


# ---------------------------------------------------------------- verification
def ecdsa_verify_rs(hash32, r, s, Q):
    """secp256k1_ecdsa_verify on already-parsed values (Q affine or None)."""
    if Q is None:
        return False
    if not (1 <= r < N and 1 <= s < N):
        return False
    if s > HALF_N:  # low-S rule (bitcoin/signature.c:185-187)
        return False
    z = int.from_bytes(hash32, "big") % N
    w = pow(s, -1, N)
    R = padd(pmul(z * w % N, G), pmul(r * w % N, Q))
    if R is None:
        return False
    return R[0] % N == r


def ecdsa_verify(hash32, sig64, pub):
    """check_signed_hash (bitcoin/signature.c:174-192) on serialized inputs:
    ok = parse_compact(sig) && parse(pub) && verify."""
    rs = sig_parse_compact(sig64)
    if rs is None:
        return False
    return ecdsa_verify_rs(hash32, rs[0], rs[1], pubkey_parse(pub))


def schnorr_verify(msg32, xonly32, sig64):
    """BIP-340 Verify == secp256k1_schnorrsig_verify(sig64, msg, 32, xonly) (bitcoin/signature.c:425-429)."""
    Pk = lift_x(int.from_bytes(xonly32, "big"))
    if Pk is None:
        return False
    r = int.from_bytes(sig64[:32], "big")
    s = int.from_bytes(sig64[32:], "big")
    if r >= P or s >= N:
        return False
    e = int.from_bytes(tagged_hash("BIP0340/challenge", sig64[:32] + xonly32 + msg32), "big") % N
    R = padd(pmul(s, G), pmul(N - e, Pk))
    if R is None or R[1] & 1 or R[0] != r:
        return False
    return True


# ---------------------------------------------------------------- signing (test-vector generation only)
def pubkey_create(d):
    return pmul(d, G)


def ser33(pt):
    return bytes([2 + (pt[1] & 1)]) + pt[0].to_bytes(32, "big")


def ser65(pt):
    return b"\x04" + pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def ecdsa_sign(hash32, d, k):
    """Textbook ECDSA with caller-chosen nonce, normalised to low-S.  -> 64-byte compact."""
    R = pmul(k, G)
    r = R[0] % N
    s = pow(k, -1, N) * (int.from_bytes(hash32, "big") + r * d) % N
    assert r and s
    if s > HALF_N:
        s = N - s
    return r.to_bytes(32, "big") + s.to_bytes(32, "big")


def schnorr_sign(msg32, d, aux32=b"\x00" * 32):
    """BIP-340 default signing."""
    Pk = pmul(d, G)
    if Pk[1] & 1:
        d = N - d
    px = Pk[0].to_bytes(32, "big")
    t = (d ^ int.from_bytes(tagged_hash("BIP0340/aux", aux32), "big")).to_bytes(32, "big")
    k0 = int.from_bytes(tagged_hash("BIP0340/nonce", t + px + msg32), "big") % N
    assert k0
    R = pmul(k0, G)
    k = N - k0 if R[1] & 1 else k0
    rx = R[0].to_bytes(32, "big")
    e = int.from_bytes(tagged_hash("BIP0340/challenge", rx + px + msg32), "big") % N
    return rx + ((k + e * d) % N).to_bytes(32, "big")


# ---------------------------------------------------------------- gossip veneer (gossipd/sigcheck.c)
def _check_nodeid(h, sig64, id33):
    return ecdsa_verify(h, sig64, id33)


def sigcheck_channel_announcement(msg):
    """gossipd/sigcheck.c:45-115 on the raw wire message.  Returns 0 = OK, or the 1-based
    index of the first bad signature (1 node_signature_1, 2 node_signature_2,
    3 bitcoin_signature_1, 4 bitcoin_signature_2), or -1 if the message is malformed
    (what fromwire_channel_announcement would have rejected before sigcheck runs)."""
    if len(msg) < 260 or msg[0:2] != b"\x01\x00":
        return -1
    flen = int.from_bytes(msg[258:260], "big")
    koff = 260 + flen + 32 + 8
    if len(msg) < koff + 4 * 33:
        return -1
    sigs = [msg[2 + 64 * i:66 + 64 * i] for i in range(4)]
    keys = [msg[koff + 33 * i:koff + 33 * i + 33] for i in range(4)]
    # fromwire: compact-sig range failure / invalid bitcoin_key => malformed (wire/fromwire.c:188-199, bitcoin/pubkey.c:102-113)
    if any(sig_parse_compact(s) is None for s in sigs):
        return -1
    if pubkey_parse(keys[2]) is None or pubkey_parse(keys[3]) is None:
        return -1
    h = sha256d(msg[258:])
    for i in range(4):
        if not _check_nodeid(h, sigs[i], keys[i]):
            return i + 1
    return 0


def bigsize_read(b, pos):
    """BigSize (BOLT #1, common/bigsize.c:53-104): (value, new position) or None when truncated / not minimal"""
    if pos >= len(b):
        return None
    t = b[pos]
    if t < 0xFD:
        return t, pos + 1
    width, floor = {0xFD: (2, 0xFD), 0xFE: (4, 1 << 16), 0xFF: (8, 1 << 32)}[t]
    if pos + 1 + width > len(b):
        return None
    v = int.from_bytes(b[pos + 1:pos + 1 + width], "big")
    return None if v < floor else (v, pos + 1 + width)


def node_ann_tlvs_ok(b):
    """the tlv stream that ends a node_announcement (wire/peer_wire.csv:367-369) under fromwire_tlv's rules
    (wire/tlvstream.c:144-300); known record 1 = lease_rates: u16 u16 u16 u32 tu32 (10..14 bytes, minimal tu32)"""
    pos, prev = 0, None
    while pos < len(b):
        r = bigsize_read(b, pos)
        if r is None:
            return False
        typ, pos = r
        if prev is not None and typ <= prev:
            return False
        prev = typ
        if typ != 1 and typ % 2 == 0:
            return False
        r = bigsize_read(b, pos)
        if r is None:
            return False
        ln, pos = r
        if ln > len(b) - pos:
            return False
        if typ == 1 and (not 10 <= ln <= 14 or (ln > 10 and b[pos + 10] == 0)):
            return False
        pos += ln
    return True


def sigcheck_channel_update(msg, node_id33):
    """gossipd/sigcheck.c:9-43.  0 = OK, 1 = 'Bad signature', -1 = malformed (fromwire_channel_update needs all
    138 bytes of fixed fields, wire/peer_wire.csv:370-381)."""
    if len(msg) < 138 or msg[0:2] != b"\x01\x02":
        return -1
    if sig_parse_compact(msg[2:66]) is None:
        return -1
    return 0 if _check_nodeid(sha256d(msg[66:]), msg[2:66], node_id33) else 1


def sigcheck_node_announcement(msg, node_id33=None):
    """gossipd/sigcheck.c:118-164.  node_id defaults to the one embedded in the message.  -1 = what
    fromwire_node_announcement rejects (truncated fixed part / addresses, bad tlv stream)."""
    if len(msg) < 68 or msg[0:2] != b"\x01\x01":
        return -1
    if sig_parse_compact(msg[2:66]) is None:
        return -1
    flen = int.from_bytes(msg[66:68], "big")
    off = 68 + flen + 4
    if len(msg) < off + 70:
        return -1
    addrlen = int.from_bytes(msg[off + 68:off + 70], "big")
    if len(msg) < off + 70 + addrlen or not node_ann_tlvs_ok(msg[off + 70 + addrlen:]):
        return -1
    if node_id33 is None:
        node_id33 = msg[off:off + 33]
    return 0 if _check_nodeid(sha256d(msg[66:]), msg[2:66], node_id33) else 1


# ---------------------------------------------------------------- BIP143 (for check_tx_sig KATs)
def _varint(n):
    if n < 0xFD:
        return bytes([n])
    if n <= 0xFFFF:
        return b"\xfd" + n.to_bytes(2, "little")
    if n <= 0xFFFFFFFF:
        return b"\xfe" + n.to_bytes(4, "little")
    return b"\xff" + n.to_bytes(8, "little")


def bip143_sighash(version, inputs, outputs, locktime, in_idx, script, amount, sighash_type):
    """inputs: [(txid32_le_bytes, vout, sequence)], outputs: [(amount, spk)].
    libwally wally_tx_get_btc_signature_hash(..., WALLY_TX_FLAG_USE_WITNESS) as called at
    bitcoin/signature.c:145-148 (check_tx_sig lets only SIGHASH_ALL and SINGLE|ANYONECANPAY through, signature.h:38-41; the other
    BIP143 modes are modelled so that the device code can be compared on every type)."""
    acp = bool(sighash_type & 0x80)
    single, none = (sighash_type & 0x1F) == 3, (sighash_type & 0x1F) == 2          # BIP143: the low five bits select the output mode
    zero = b"\x00" * 32
    hp = zero if acp else sha256d(b"".join(t + v.to_bytes(4, "little") for t, v, _ in inputs))
    hs = zero if (acp or single or none) else sha256d(b"".join(s.to_bytes(4, "little") for _, _, s in inputs))

    def ser_out(o):
        return o[0].to_bytes(8, "little") + _varint(len(o[1])) + o[1]
    if single:
        ho = sha256d(ser_out(outputs[in_idx])) if in_idx < len(outputs) else zero
    elif none:
        ho = zero
    else:
        ho = sha256d(b"".join(ser_out(o) for o in outputs))
    t, v, s = inputs[in_idx]
    pre = (version.to_bytes(4, "little") + hp + hs + t + v.to_bytes(4, "little")
           + _varint(len(script)) + script + amount.to_bytes(8, "little")
           + s.to_bytes(4, "little") + ho + locktime.to_bytes(4, "little")
           + sighash_type.to_bytes(4, "little"))
    return sha256d(pre), pre


def grind_htlc_tx_fee(preimage, outputs, input_sat, weight, min_feerate, max_feerate, sig64, sighash_type, has_witness, pub33,
                      verify=None):
    """grind_htlc_tx_fee(), onchaind/onchaind.c:388-438, on serialised inputs: `preimage` is the BIP143 preimage of the
    transaction as it stands (hashOutputs sits 40 bytes before its end, bitcoin/signature.c:120-151), `outputs` the serialised
    outputs hashOutputs covers (amount of output 0 first).  Returns (feerate, fee) of the first feerate whose fee verifies,
    else None.  fee = amount_tx_fee(feerate, weight) (common/amount.c:698-707); equal consecutive fees are tried once
    (:420-424); a fee above the input amount ends the loop (:425-426); the check is check_tx_sig (:430-432) with its
    sighash-type gate (bitcoin/signature.c:206-211).  `verify(hash32, sig64, pub33)` defaults to ecdsa_verify."""
    verify = verify or ecdsa_verify
    prev = None
    for rate in range(min_feerate, max_feerate + 1):
        fee = rate * weight // 1000
        if fee == prev:
            continue
        prev = fee
        if fee > input_sat:
            break
        if not (sighash_type == 1 or (sighash_type == 0x83 and has_witness)):
            continue
        outs = (input_sat - fee).to_bytes(8, "little") + bytes(outputs[8:])
        pre = bytes(preimage[:-40]) + sha256d(outs) + bytes(preimage[-8:])
        if verify(sha256d(pre), sig64, pub33):
            return rate, fee
    return None


def ecdsa_recover(hash32, sig64, recid):
    """secp256k1_ecdsa_recoverable_signature_parse_compact + secp256k1_ecdsa_recover as called at common/bolt11.c:1021-1046
    and lightningd/signmessage.c:193 (SEC1 4.1.6 with the library's failure rules): the public key point, or None.
    Fails when r or s >= n (parse), recid not in 0..3, r or s = 0, recid & 2 and r >= p - n, no point with that x,
    or the result is infinity.  No low-S requirement."""
    r = int.from_bytes(sig64[:32], "big")
    s = int.from_bytes(sig64[32:], "big")
    if r >= N or s >= N or not 0 <= recid <= 3:
        return None
    if r == 0 or s == 0:
        return None
    x = r
    if recid & 2:
        if r >= P - N:
            return None
        x = r + N
    R = lift_x(x)                       # even y
    if R is None:
        return None
    if recid & 1:
        R = pneg(R)
    z = int.from_bytes(hash32, "big") % N
    rinv = pow(r, -1, N)
    return padd(pmul(s * rinv % N, R), pmul((-z * rinv) % N, G))


# ---- BOLT #12 signatures: merkle_tlv() + sighash_from_merkle() (common/bolt12_merkle.c:48-318) in front of BIP-340
def bigsize(v):
    """common/bigsize.c bigsize_put"""
    if v < 0xfd:
        return bytes([v])
    if v <= 0xffff:
        return b"\xfd" + v.to_bytes(2, "big")
    if v <= 0xffffffff:
        return b"\xfe" + v.to_bytes(4, "big")
    return b"\xff" + v.to_bytes(8, "big")


def tlv_stream_parse(b):
    """the generic rules of fromwire_tlv (wire/tlvstream.c:144-300) with every type accepted: BigSize type and length minimally
    encoded, lengths inside the stream, types strictly increasing -> [(type, value)] or None"""
    out, pos, prev = [], 0, None
    while pos < len(b):
        r = bigsize_read(b, pos)
        if r is None:
            return None
        t, pos = r
        if prev is not None and t <= prev:
            return None
        prev = t
        r = bigsize_read(b, pos)
        if r is None:
            return None
        ln, pos = r
        if ln > len(b) - pos:
            return None
        out.append((t, b[pos:pos + ln]))
        pos += ln
    return out


def bolt12_H(tag, msg):
    """BOLT #12: H(tag, msg) = SHA256(SHA256(tag) || SHA256(tag) || msg) (bolt12_merkle.c:48-58)"""
    t = hashlib.sha256(tag).digest()
    return hashlib.sha256(t + t + msg).digest()


def bolt12_merkle(fields):
    """fields: [(type, value)] in stream order; signature fields (240..1000) are skipped; the nonce tag comes from the first
    field that becomes a leaf's predecessor exactly as merkle_tlv_full_() does it (:262-286); the tree is the reference's
    oversized power-of-two with absent nodes passed through (:186-225, :293-296)"""
    ser = lambda t, v: bigsize(t) + bigsize(len(v)) + v
    pair = lambda a, b: bolt12_H(b"LnBranch", min(a, b) + max(a, b))
    leaves, first = [], None
    for t, v in fields:
        if not leaves:
            first = ser(t, v)
        if 240 <= t <= 1000:
            continue
        leaves.append(pair(bolt12_H(b"LnLeaf", ser(t, v)), bolt12_H(b"LnNonce" + first, bigsize(t))))
    if not leaves:
        return None

    def rec(arr):
        if len(arr) == 1:
            return arr[0]
        left, right = rec(arr[:len(arr) // 2]), rec(arr[len(arr) // 2:])
        return left if right is None else pair(left, right)
    size = 1 << len(leaves).bit_length()          # 1 << ilog64(n)
    return rec(leaves + [None] * (size - len(leaves)))


def bolt12_sighash(messagename, fieldname, merkle):
    """sighash_from_merkle (bolt12_merkle.c:308-318) with bip340_sighash_init's three-part tag (bitcoin/signature.c:389-405)"""
    return bolt12_H(b"lightning" + messagename + fieldname, merkle)


def bolt12_check_signature(tlv_stream, messagename, fieldname, key33, sig64):
    """bolt12_check_signature (common/bolt12.c:80-92) on a serialised TLV stream; check_schnorr_sig drops the key's parity byte"""
    fields = tlv_stream_parse(tlv_stream)
    if fields is None:
        return False
    m = bolt12_merkle(fields)
    if m is None:
        return False
    return schnorr_verify(bolt12_sighash(messagename, fieldname, m), key33[1:], sig64)

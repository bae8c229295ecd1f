"""Generates proqa_amd/csrc/wordpiece_tables.inc: what BertTokenizer's normaliser and pre-tokeniser do to every code point
of the Basic Multilingual Plane, taken from the `tokenizers` library itself (BertNormalizer / BertPreTokenizer probed one
code point at a time), so that csrc/wordpiece.cpp can tokenise non-ASCII text natively with the same ids.

Per code point c (the text "x" + c + "y" through the normaliser):
    removed      clean_text drops it (NUL, U+FFFD, control / format / private-use characters)
    whitespace   clean_text turns it into a space
    cjk          handle_chinese_chars puts spaces around it
    punct        the pre-tokeniser makes it a token of its own
    mapped       uncased models only: NFD + strip Mn + lowercase changes it (the replacement sequence goes to the pool;
                 an empty sequence = a nonspacing mark that is stripped)
    hangul       uncased models only: a Hangul syllable, decomposed algorithmically (checked against the library here)
    fallback     a mark with a non-zero combining class that survives the stripping (canonical REORDERING could move it
                 across its neighbours: such texts go to the reference tokenizer)
Cased models (no NFD, no lowercase) must leave every kept character alone: asserted.

usage: python scripts/gen_wordpiece_tables.py [--check]     (--check: compare with the committed file, write nothing)
"""
import os
import random
import sys
import unicodedata

from tokenizers import normalizers, pre_tokenizers
import tokenizers

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proqa_amd", "csrc", "wordpiece_tables.inc")
REMOVED, SPACE, CJK, PUNCT, MAPPED, FALLBACK, HANGUL = 1, 2, 4, 8, 16, 32, 64

lower = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=True)
cased = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None, lowercase=False)
pre = pre_tokenizers.BertPreTokenizer()


def hangul_nfd(cp):
    s = cp - 0xAC00
    out = [0x1100 + s // 588, 0x1161 + (s % 588) // 28]
    if s % 28:
        out.append(0x11A7 + s % 28)
    return out


flags = [0] * 0x10000
mapping = {}
for cp in range(0x10000):
    if 0xD800 <= cp <= 0xDFFF:
        flags[cp] = FALLBACK          # not encodable; never reached from valid UTF-8
        continue
    c = chr(cp)
    n = lower.normalize_str("x" + c + "y")
    nc = cased.normalize_str("x" + c + "y")
    assert n[0] == "x" and n[-1] == "y" and nc[0] == "x" and nc[-1] == "y", hex(cp)
    body, body_c = n[1:-1], nc[1:-1]
    f = 0
    if body_c == "":
        f |= REMOVED
        assert body == "", hex(cp)
    elif body_c == " ":
        f |= SPACE
        assert body == " ", hex(cp)
    else:
        if body_c.startswith(" ") and body_c.endswith(" ") and len(body_c) >= 3:
            f |= CJK
            body_c = body_c[1:-1]
            assert body.startswith(" ") and body.endswith(" "), hex(cp)
            body = body[1:-1]
        assert body_c == c, ("a cased model changes", hex(cp), body_c)
        pieces = [p for p, _ in pre.pre_tokenize_str("x" + c + "y")]
        if pieces == ["x", c, "y"]:
            f |= PUNCT
        else:
            assert pieces == ["x" + c + "y"], (hex(cp), pieces)
        if body != c:
            if 0xAC00 <= cp <= 0xD7A3:
                assert [ord(ch) for ch in body] == hangul_nfd(cp), hex(cp)
                f |= HANGUL
            else:
                if all(ord(ch) < 0x10000 for ch in body):
                    f |= MAPPED
                    mapping[cp] = [ord(ch) for ch in body]
                else:
                    f |= FALLBACK          # decomposes to a character outside the BMP (a few compatibility ideographs)
        if unicodedata.combining(c) != 0 and unicodedata.category(c) != "Mn":
            f |= FALLBACK
    flags[cp] = f

# every output of a mapping is itself stable (no second pass needed) and keeps / drops nothing by itself
for cp, outs in mapping.items():
    for o in outs:
        assert not flags[o] & (REMOVED | SPACE | MAPPED | HANGUL), (hex(cp), hex(o))
# context check on random sequences of "interesting" characters: per-character tables == the library on the whole string
rng = random.Random(5)
interesting = [cp for cp in range(0x80, 0x10000) if flags[cp] & (MAPPED | HANGUL | CJK | PUNCT) and not flags[cp] & FALLBACK]
marks = [cp for cp in range(0x300, 0x10000) if unicodedata.combining(chr(cp)) and not flags[cp] & FALLBACK]


def emulate(text):
    out = []
    for ch in text:
        cp = ord(ch)
        f = flags[cp]
        if f & REMOVED:
            continue
        if f & SPACE:
            out.append(" ")
            continue
        seq = mapping[cp] if f & MAPPED else hangul_nfd(cp) if f & HANGUL else [cp]
        s = "".join(chr(o) for o in seq)
        out.append(" " + s + " " if f & CJK else s)
    return "".join(out)


bad = 0
for _ in range(20000):
    text = "".join(chr(rng.choice(interesting if rng.random() < 0.6 else marks if rng.random() < 0.7 else range(0x20, 0x7F)))
                   for _ in range(rng.randint(1, 12)))
    if emulate(text) != lower.normalize_str(text):
        bad += 1
        if bad < 5:
            print("context mismatch:", [hex(ord(c)) for c in text], file=sys.stderr)
assert bad == 0, f"{bad} context-dependent strings: extend the fallback class"

ranges = []
for cp in range(0x10000):
    if ranges and ranges[-1][2] == flags[cp] and ranges[-1][1] == cp - 1:
        ranges[-1][1] = cp
    else:
        ranges.append([cp, cp, flags[cp]])
keys = sorted(mapping)
offs, pool = [0], []
for k in keys:
    pool.extend(mapping[k])
    offs.append(len(pool))

lines = [
    "// GENERATED by scripts/gen_wordpiece_tables.py from tokenizers %s (BertNormalizer / BertPreTokenizer probed per code point;" % tokenizers.__version__,
    "// Python unicodedata %s for the combining classes of the fallback set).  Do not edit." % unicodedata.unidata_version,
    "// flags: 1 removed, 2 whitespace, 4 cjk (spaced), 8 punctuation, 16 mapped (uncased), 32 fallback, 64 hangul syllable (uncased)",
    "static const struct { uint16_t first, last; uint8_t flags; } kWpClassRanges[] = {",
]
row = []
for a, b, f in ranges:
    if f == 0:
        continue
    row.append("{0x%04X,0x%04X,%d}" % (a, b, f))
    if len(row) == 8:
        lines.append("  " + ",".join(row) + ",")
        row = []
if row:
    lines.append("  " + ",".join(row) + ",")
lines.append("};")


def emit_array(ctype, name, vals, fmt, per=16):
    lines.append("static const %s %s[] = {" % (ctype, name))
    for i in range(0, len(vals), per):
        lines.append("  " + ",".join(fmt % v for v in vals[i:i + per]) + ",")
    lines.append("};")


emit_array("uint16_t", "kWpMapKeys", keys, "0x%04X")
emit_array("uint32_t", "kWpMapOffsets", offs, "%d", 20)
emit_array("uint16_t", "kWpMapPool", pool, "0x%04X")
text = "\n".join(lines) + "\n"
if "--check" in sys.argv:
    same = os.path.exists(OUT) and open(OUT).read() == text
    print("tables are up to date" if same else "tables DIFFER from the committed file")
    sys.exit(0 if same else 1)
with open(OUT, "w") as f:
    f.write(text)
print(f"{OUT}: {len(ranges)} class ranges ({sum(1 for r in ranges if r[2])} non-zero), {len(keys)} mapped code points, pool of {len(pool)}; "
      f"fallback code points: {sum(1 for cp in range(0x10000) if flags[cp] & FALLBACK and not 0xD800 <= cp <= 0xDFFF)}")

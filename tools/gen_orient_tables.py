"""Constant tables of the keypoint orientation (overlap stage): the 7x7 Gaussian weights
exp(-(i^2+j^2)/(2*2.5^2)) and, for the 42 sector starts a_k = 0.15 k rad, the unit vectors of a_k and a_k + pi/3.
The same literal text is pasted into oracle/uwip_oracle_overlap.c and uwimageproc_amd/csrc/overlap.hip (between the
ORIENT-TABLES markers) so that both sides hold bit-identical floats:   python tools/gen_orient_tables.py --write"""
import os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lit(v):
    s = "%.9g" % float(np.float32(v))
    return s + ("f" if ("." in s or "e" in s) else ".0f")


def text(prefix):
    out = [f"static const float {prefix}GAUSS25[7][7] = {{"]
    for i in range(7):
        out.append("    {" + ", ".join(lit(np.exp(-(i * i + j * j) / (2 * 2.5 * 2.5))) for j in range(7)) + "},")
    out.append("};")
    out.append(f"/* sector k: [a_k, a_k + pi/3), a_k = 0.15 k; {{cos a_k, sin a_k, cos(a_k + pi/3), sin(a_k + pi/3)}} */")
    out.append(f"static const float {prefix}SECTOR[42][4] = {{")
    for k in range(42):
        a = k * 0.15
        b = a + np.pi / 3
        out.append("    {%s, %s, %s, %s}," % (lit(np.cos(a)), lit(np.sin(a)), lit(np.cos(b)), lit(np.sin(b))))
    out.append("};")
    if prefix == "D_":
        disc = [(i, j) for i in range(-6, 7) for j in range(-6, 7) if i * i + j * j < 36]
        assert len(disc) == 109
        out.append("/* the 109 lattice points of the radius-6 disc in the oracle's loop order (i = x offset outer, j = y offset inner) */")
        out.append("static const signed char D_DISC[109][2] = {")
        for k in range(0, 109, 12):
            out.append("    " + " ".join("{%d, %d}," % ij for ij in disc[k:k + 12]))
        out.append("};")
    return "\n".join(out)


if __name__ == "__main__":
    targets = [("oracle/uwip_oracle_overlap.c", "OV_", ""), ("uwimageproc_amd/csrc/overlap.hip", "D_", "__device__ ")]
    for path, prefix, qual in targets:
        t = text(prefix)
        if qual:
            t = t.replace("static const", f"static {qual}const")
        if "--write" in sys.argv:
            p = os.path.join(ROOT, path)
            s = open(p).read()
            s2 = re.sub(r"(/\* ORIENT-TABLES-BEGIN[^\n]*\*/\n).*?(/\* ORIENT-TABLES-END \*/)", lambda m: m.group(1) + t + "\n" + m.group(2), s, flags=re.S)
            assert s2 != s or t in s, f"markers not found in {path}"
            open(p, "w").write(s2)
        else:
            print(t)

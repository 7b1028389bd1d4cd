"""tests/golden/wolf_1k.ply: the first 1000 Gaussians of the reference's own sample
source/Gradio/favorites/wolf.spz, decoded by the reference's own codec (oracle/_ref/
splat_converter, built by oracle/Makefile from the sources in /root/reference).  A real-world
PLY in exactly the byte layout the reference's tools write and read."""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/source/Gradio/favorites/wolf.spz"
with tempfile.TemporaryDirectory() as td:
    shutil.copy(src, os.path.join(td, "wolf.spz"))
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "splat_converter"), os.path.join(td, "wolf.spz")], check=True)
    raw = open(os.path.join(td, "wolf.ply"), "rb").read()
end = raw.index(b"end_header\n") + len(b"end_header\n")
header = raw[:end].decode()
n = int([l for l in header.splitlines() if l.startswith("element vertex")][0].split()[2])
nprop = sum(1 for l in header.splitlines() if l.startswith("property"))
keep = 1000
header = header.replace(f"element vertex {n}", f"element vertex {keep}")
out = os.path.join(ROOT, "tests", "golden", "wolf_1k.ply")
open(out, "wb").write(header.encode() + raw[end:end + keep * nprop * 4])
print("wrote", out, nprop, "properties")

"""The inserter's straight-line asm in lz_links is generated (tools/gen_links_block.py simulates the LDS queue to count its
waits): the text in the kernel source must be what the generator prints for the depth the kernel's comment names."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_links_block_in_source_is_the_generated_one():
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_links_block.py"), "2", "3"], capture_output=True, text=True, check=True)
    lines = [l.strip() for l in gen.stdout.splitlines() if l.strip()]
    assert len(lines) > 100
    src = open(os.path.join(ROOT, "parallel-data-compression-and-decompression_amd", "csrc", "zwz_kernels.hip")).read()
    block = "\n".join(l.strip() for l in src.splitlines())
    assert "\n".join(lines) in block, "zwz_kernels.hip's inserter block differs from tools/gen_links_block.py 2 3"
    clobbers = gen.stderr.strip().replace("// clobbers: ", "")
    assert clobbers in src


def test_links_block_waits_never_exceed_the_queue():
    # every depth the generator accepts keeps its counted waits within the hardware counter (lgkmcnt is 4 bits)
    for ahead, behind in ((1, 2), (2, 3), (3, 4)):
        gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_links_block.py"), str(ahead), str(behind)], capture_output=True, text=True, check=True)
        waits = [int(l.split("lgkmcnt(")[1].split(")")[0]) for l in gen.stdout.splitlines() if "lgkmcnt(" in l]
        assert waits and max(waits) <= 15

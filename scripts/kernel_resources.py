"""Print the register / LDS / scratch footprint of the kernels in a class library (code-object metadata).
   python scripts/kernel_resources.py path/to/lib<cls>_<tag>.so"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def resources(so):
    with tempfile.TemporaryDirectory() as d:
        co, fb = d + "/k.co", d + "/k.fatbin"
        subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fb], check=True)
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True,
                       capture_output=True)
        txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
        out[name] = dict(agpr=int(blk.split()[0]), vgpr=g("vgpr_count"), sgpr=g("sgpr_count"),
                         lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"),
                         vgpr_spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"))
    return out


if __name__ == "__main__":
    for so in sys.argv[1:]:
        for k, v in resources(so).items():
            print(so.split("/")[-1], k, v)

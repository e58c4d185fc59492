import sys, os, time, cProfile, pstats, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import mirge3_amd
from mirge3_amd import synth
from mirge3_amd.seqio import write_fasta, index_basename
from mirge3_amd import cli
tmp = tempfile.mkdtemp()
sl = synth.make_libraries(seed=3, scale="small")
idx = os.path.join(tmp, "Libs", "human", "index.Libs"); ann = os.path.join(tmp, "Libs", "human", "annotation.Libs")
os.makedirs(idx); os.makedirs(ann)
for k, lib in sl.libs.items():
    write_fasta(os.path.join(idx, index_basename("human", k, "miRBase") + ".fa"), lib)
open(os.path.join(ann, "human_merges_miRBase.csv"), "w").write("".join(",".join(r) + "\n" for r in sl.merges))
reads = synth.make_reads_chunked(sl, 3_000_000, seed=4).to_list()
fq = os.path.join(tmp, "S1.fastq")
with open(fq, "w") as fh:
    fh.write("".join(f"@r\n{r}\n+\n{'I'*len(r)}\n" for r in reads))
print("fastq MB", os.path.getsize(fq) / 1e6, flush=True)
argv = ["-s", fq, "-lib", os.path.join(tmp, "Libs"), "-on", "human", "-db", "miRBase", "-o", tmp, "-dn", "out", "-shh"]
t = time.time()
pr = cProfile.Profile(); pr.enable()
cli.main(argv)
pr.disable()
print("cli wall", round(time.time() - t, 2), "s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

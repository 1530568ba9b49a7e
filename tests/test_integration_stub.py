"""INTEGRATION.md section 2 promises "this is the whole stub": the ctypes block is cut out of the
document and executed verbatim on the GPU, then compared with the Engine path."""
import os
import re

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_ctypes_stub_of_integration_md_runs_verbatim(native_lib, tmp_path):
    from mirge_amd import _native, annotate, synth
    from mirge_amd.engine import Engine
    from mirge_amd.index import FmIndex
    from tests.golden.make_golden import SHAPES
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- stub:ctypes.*?-->\s*```python\n(.*?)\n```", text, re.S)
    assert m, "the marked ctypes block is gone from INTEGRATION.md"
    libs = synth.SynthLibraries(seed=123, scale=1.0, n_paralogs=6, n_snp=8, shapes=SHAPES)
    prefix = {}
    for key, (names, seqs) in libs.libs.items():
        prefix[key] = str(tmp_path / key)
        FmIndex.build(names, seqs).save(prefix[key] + ".mrgfm")
    reads = list(dict.fromkeys(synth.codes_to_str(c) for c in synth.synth_reads(libs, 800, seed=3, zipf_s=1.3)))
    long_mrna = max(libs.libs["mrna"][1], key=len)
    reads += ["ACGTNACGTTAGCATCGATCGA", "A" * 140, "A" * 300, libs.libs["mrna"][1][0][7:180], long_mrna[1:299],
              long_mrna[4:150] + "N" + long_mrna[151:290]]
    make = lambda: {s: {"quant": [1], "annot": [0] + [""] * 9, "length": len(s)} for s in reads}
    seq_dic = make()
    env = dict(LIB=_native.LIB_PATH, index_prefix=prefix, seqDic=seq_dic)
    exec(compile(m.group(1), "INTEGRATION.md:stub", "exec"), env)
    # the same through the Python-level swap of section 1
    want = make()
    eng = Engine(0)
    annotate.runAnnotationPipeline(eng, want, "1", False, [], str(tmp_path), {"annotStats": []}, prefix["mirna"],
                                   prefix["hairpin"], prefix["mature_trna"], prefix["pre_trna"], prefix["snorna"],
                                   prefix["rrna"], prefix["ncrna_others"], prefix["mrna"], False, None, False, None,
                                   None, "miRBase", False, None, None, ["s"])
    assert {s: r["annot"] for s, r in seq_dic.items()} == {s: r["annot"] for s, r in want.items()}
    assert sum(r["annot"][0] for r in seq_dic.values()) > len(reads) // 2
    assert seq_dic["A" * 140]["annot"][0] == 0 and seq_dic["A" * 300]["annot"][0] == 0
    assert seq_dic[libs.libs["mrna"][1][0][7:180]]["annot"][0] == 1   # a 173-nt read, aligned (eight words)
    assert seq_dic[long_mrna[1:299]]["annot"][8] != "" and seq_dic[long_mrna[4:150] + "N" + long_mrna[151:290]]["annot"][8] != ""

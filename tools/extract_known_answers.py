#!/usr/bin/env python3
"""Dev-time tool: collect the known-answer LITERALS (inputs + expected values) of the reference's unit tests
into tests/golden/known_answers.json.  Only data is stored (sequences, counts, expected numbers), each with the
test file:line it came from.  Needs /root/reference; the JSON it writes is committed and is what the tests read."""
import json, re, pathlib

T = pathlib.Path("/root/reference/src/test/scala")
OUT = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden" / "known_answers.json"

def seqs(text):
    return re.findall(r'"([ACGT]{23})"', text)

def main():
    ka = {}
    # ---- bitcoding/BitEncodingTest.scala ----
    ka["mismatch_cases"] = [
        # enzyme index, s1, c1, s2, c2, expected, source
        [2, "AAAAACCCCCGGGGGTTTTAGGG", 1000, "AAAAACCCCCGGGGGTTTTAGGG", 1000, 0, "BitEncodingTest.scala:79-86"],
        [2, "AAAAACCCCCGGGGGTTTTAGGG", 1000, "AAAAACCCCCGGGGGTTTTAGGG", 1001, 0, "BitEncodingTest.scala:88-99"],
        [2, "AAAAACCCCCGGGGGTTTTAGGG", 1000, "TAAAACCCCCGGGGGTTTTAGGG", 1001, 1, "BitEncodingTest.scala:101-112"],
        [2, "AAAAACCCCCGGGGGTTTTAGGG", 1000, "TTTTTTTTTTAAAAAGGGGGGGG", 1001, 20, "BitEncodingTest.scala:114-125"],
        [2, "AAAAACCCCCGGGGGAAAATAGG", 1000, "AAAAACCCCCGGGGGTTTTAGGG", 1001, 5, "BitEncodingTest.scala:127-138"],
        [2, "AAAAACCCCCGGGGGAAAATAAG", 1000, "AAAAACCCCCGGGGGTTTTAGGG", 1001, 5, "BitEncodingTest.scala:140-151"],
        [3, "GAGTCCGAGCAGAAGAAGAAGGG", 1, "GAATCATAGCAGAAGATGAAAGG", 1001, 4, "BitEncodingTest.scala:296-307"],
    ]
    ka["bin_cases"] = [
        # enzyme, guide, bin, expected mismatchBin, source
        [2, "AAAAACCCCCGGGGGTTTTAGGG", "AAAAA", 0, "BitEncodingTest.scala:236-250"],
        [2, "TTAATCCCCCGGGGGTTTTAGGG", "TTTTT", 2, "BitEncodingTest.scala:252-264"],
        [2, "AAAAAAAAACGGGGGTTTTAGGG", "AAAAAAAAA", 0, "BitEncodingTest.scala:266-278"],
        [3, "GAGTCCGAGCAGAAGAAGAAGGG", "GAGTCCG", 0, "BitEncodingTest.scala:310-319"],
        [1, "TTTTCGAGCAGAAGAAGAAGGGAC", "CGAGCAG", 0, "BitEncodingTest.scala:321-330"],
        [1, "TTTTCGAGCAGAAGAAGAAGGGAC", "CAAGCAG", 1, "BitEncodingTest.scala:332-333"],
        [1, "TTTTCGAGCAGAAGAAGAAGGGAC", "AGAGCAA", 2, "BitEncodingTest.scala:335-336"],
        [3, "GGCTCCGAGCAGAAGAAGAAGGG", "GAGTCCG", 2, "BitEncodingTest.scala:338-347"],
        [3, "GGCTCCGAGCAGAAGAAGAAGGG", "AAAAAAA", 7, "BitEncodingTest.scala:350-359"],
    ]
    ka["roundtrip_case"] = ["AAAAACCCCCGGGGGTTTTAGGG", 1000, "BitEncodingTest.scala:20-30"]
    # ---- utils/UtilsTest.scala:38-46 ----
    ka["long_bytes_case"] = {"longs": [0x0BCDEFABCDEFABCD, 0, 1], "byte7": 0x0B, "byte16": 0x01, "source": "UtilsTest.scala:38-46"}
    # ---- scoring/Doench2016CFDScoreTest.scala ----
    cfd = (T / "scoring/Doench2016CFDScoreTest.scala").read_text()
    blocks = cfd.split('"Doench2016CFDScore" should')
    ka["cfd_pairs"] = {"guide": "GACTTGCATCCGAAGCCGGT", "tol": 1e-3, "source": "Doench2016CFDScoreTest.scala:32-40",
                       "cases": [[m[0], float(m[1])] for m in re.findall(r'scoreCFD\(guide,"([ACGT]{20})"\)\) should be\(([0-9.]+)', cfd)]}
    g1 = seqs(blocks[1]); g3 = seqs(blocks[3]); g4 = seqs(blocks[4])
    ka["cfd_guides"] = [
        {"guide": "CGCGCGGCCCCAGTTCTGCGCAG", "hits": g1[:1], "maxOT_printed": 0.0, "tol": 1e-3, "source": "Doench2016CFDScoreTest.scala:19-29"},
        {"guide": "AAAAGGGTTTGGGATATAGCTGG", "hits": g3[:19], "maxOT_printed": 0.5238095242619047, "tol": 1e-3, "source": "Doench2016CFDScoreTest.scala:43-56"},
        {"guide": "CGCGCGGCCCCAGTTCTGCGCAG", "hits": g4[:87], "maxOT_printed": 0.30252100830756307, "tol": 1e-3, "source": "Doench2016CFDScoreTest.scala:58-84"},
    ]
    assert len(ka["cfd_pairs"]["cases"]) == 5 and len(ka["cfd_guides"][1]["hits"]) == 19 and len(ka["cfd_guides"][2]["hits"]) == 87
    # ---- scoring/CrisprMitEduOffTargetTest.scala ----
    mit = (T / "scoring/CrisprMitEduOffTargetTest.scala").read_text()
    head, tail = mit.split('"CrisprMitEduOffTargetTest" should', 1)
    hs = seqs(head)
    ka["hsu_guide"] = {"guide": hs[0], "hits": hs[1:], "expected": 96.0, "tol": 1.0, "source": "CrisprMitEduOffTargetTest.scala:15-58"}
    assert len(hs) == 31
    ka["hsu_pair"] = {"guide": "TTGTTTCCAGGTCAATGTGACGG", "ot": "TTGTCTTCAAGTCAATATGATGG", "expected": 0.36403873, "tol": 0.1,
                      "source": "CrisprMitEduOffTargetTest.scala:61-70"}
    # ---- scoring/ClosestHitTest.scala:23-59 (mismatch placements are random there; counts/levels are the literals) ----
    ka["closest_cases"] = [
        {"guide": "GACTTGCATCCGAAGCCGGTGGG", "mm": [1], "counts": [1], "closest": "1", "count": "1", "hist": "0,1,0,0,0", "source": "ClosestHitTest.scala:23-34"},
        {"guide": "GACTTGCATCCGAAGCCGGTGGG", "mm": [1], "counts": [40], "closest": "1", "count": "40", "hist": "0,40,0,0,0", "source": "ClosestHitTest.scala:36-46"},
        {"guide": "GACTTGCATCCGAAGCCGGTGGG", "mm": [1, 1, 2, 4], "counts": [40, 30, 20, 10], "closest": "1", "count": "70", "hist": "0,70,20,0,10", "source": "ClosestHitTest.scala:49-59"},
    ]
    # ---- reference/SimpleSiteFinderTest.scala ----
    ka["site_cases"] = [
        # enzyme, flank, sequence, expected [[bases, start, forward, has_context]], source
        [3, 0, "ATTTAAAAAACCCCCAAAAAGGG", [["ATTTAAAAAACCCCCAAAAAGGG", 0, True, True]], "SimpleSiteFinderTest.scala:13-27"],
        [3, 8, "ATAATATAATTTAAAAAATTTTTAAAAAAGGAATTAAAT", [["ATTTAAAAAATTTTTAAAAAAGG", 8, True, True]], "SimpleSiteFinderTest.scala:29-43"],
        [3, 0, "CCTTAAAAAACCCCCAAAAAAAA", [["TTTTTTTTGGGGGTTTTTTAAGG", 0, False, True]], "SimpleSiteFinderTest.scala:45-56"],
        [3, 0, "AATTTAAAAAACCCCCAAAAAGGG", [["AATTTAAAAAACCCCCAAAAAGG", 0, True, True], ["ATTTAAAAAACCCCCAAAAAGGG", 1, True, True]], "SimpleSiteFinderTest.scala:58-72"],
        [4, 0, "ATTTAAAAAACCCCCAAAAAGAG", [["ATTTAAAAAACCCCCAAAAAGAG", 0, True, True]], "SimpleSiteFinderTest.scala:74-85"],
        [4, 0, "CTTTAAAAAACCCCCAAAAAAAA", [["TTTTTTTTGGGGGTTTTTTAAAG", 0, False, True]], "SimpleSiteFinderTest.scala:87-98"],
        [2, 0, "AATTTAAAAAACCCCCAAAAAAGG", [["AATTTAAAAAACCCCCAAAAAAG", 0, True, True], ["ATTTAAAAAACCCCCAAAAAAGG", 1, True, True]], "SimpleSiteFinderTest.scala:99-113"],
        [3, 0, "AAATAAAAAACCCCCAAAAAGGG", [["AAATAAAAAACCCCCAAAAAGGG", 0, True, True]], "SimpleSiteFinderTest.scala:115-126"],
        [1, 0, "TTTTAATTTAAAAAACCCCCAATTT", [["TTTTAATTTAAAAAACCCCCAATT", 0, True, True], ["TTTAATTTAAAAAACCCCCAATTT", 1, True, True]], "SimpleSiteFinderTest.scala:128-142"],
        [1, 0, "TAATAATTTAAAAAACCCCCAAAAA", [["TTTTGGGGGTTTTTTAAATTATTA", 0, False, True], ["TTTTTGGGGGTTTTTTAAATTATT", 1, False, True]], "SimpleSiteFinderTest.scala:144-158"],
        [3, 1, "ATTTAAAAAACCCCCAAAAAGGG", [["ATTTAAAAAACCCCCAAAAAGGG", 0, True, False]], "SimpleSiteFinderTest.scala:161-173"],
    ]
    # ---- scoring/JoistAndSantosCRISPRiTest.scala (exact equality in the reference's tests: `should be(...)`) ----
    A20 = "A" * 20
    ka["jost_pairs"] = [
        # enzyme, target, off-target, expected factors (multiplied left to right), source
        [2, A20 + "GGG", "T" + "A" * 19 + "GGG", [1.0], "JoistAndSantosCRISPRiTest.scala:18-22"],
        [2, A20 + "GGG", "AT" + "A" * 18 + "GGG", [0.7952747759038213], "JoistAndSantosCRISPRiTest.scala:24-26"],
        [2, A20 + "GGG", "AAAATAAAATAAAAGAAAAAGGG", [0.6947382165440157, 0.31016952886752025, 0.26865890093507167], "JoistAndSantosCRISPRiTest.scala:30-34"],
        [2, A20 + "GGG", "ATAAAAAAAAAAAAAAAAATGGG", [0.7952747759038213, 0.03182081449682617], "JoistAndSantosCRISPRiTest.scala:36-38"],
    ]
    ka["jost_guides"] = [
        # enzyme, guide, hits, expected maxOT factors (empty = "0.0": nothing scored), source
        [2, A20 + "GGG", [A20 + "GGG"], [], "JoistAndSantosCRISPRiTest.scala:41-48"],
        [2, A20 + "GGG", ["AAAATAAAATAAAAGAAAAAGGG"], [0.6947382165440157, 0.31016952886752025, 0.26865890093507167], "JoistAndSantosCRISPRiTest.scala:50-57"],
        [2, A20 + "AGG", ["T" + "A" * 20 + "GG"], [1.0], "JoistAndSantosCRISPRiTest.scala:59-66"],
        [5, "A" * 19 + "GGG", ["AAATAAAATAAAAGAAAAAGGG"], [0.6947382165440157, 0.31016952886752025, 0.26865890093507167], "JoistAndSantosCRISPRiTest.scala:68-84"],
    ]
    OUT.write_text(json.dumps(ka, indent=1))
    print("wrote", OUT)

if __name__ == "__main__":
    main()

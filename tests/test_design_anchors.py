"""CPU-only: every file:line anchor of DESIGN.md's front page names the code it claims to name. The anchors point into files
that are still edited; this test makes a stale line number a failing test instead of a reader's puzzle."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "opv-cxx-demod_amd"

# (file, regular expression of the line the anchor must point at, how DESIGN.md writes the anchor: {n} = the line number)
ANCHORS = [
    ("csrc/k_frontend.hip", r"auto issue_tile = \[&\]", "`csrc/k_frontend.hip:{n}`"),
    ("csrc/k_frontend.hip", r"const int s0r = ", "`:{n}`"),
    ("csrc/k_frontend.hip", r"auto symbol_r = \[&\]", "`csrc/k_frontend.hip:{n}` (`symbol_r`)"),
    ("csrc/k_frontend.hip", r"^    for \(;;\) \{", "`csrc/k_frontend.hip:{n}` (chunk loop"),
    ("csrc/k_frontend.hip", r"void k_msk_frontend_rb\(", "`:{n},"),
    ("csrc/k_frontend.hip", r"void k_msk_frontend_rb_wg4\(", ",:{n}`"),
    ("csrc/k_frontend.hip", r"^constexpr uint32_t kTile = ", "`csrc/k_frontend.hip:{n}-"),
    ("csrc/k_frontend_x4.hip", r"void k_msk_frontend_x4_wg4\(", "`csrc/k_frontend_x4.hip:{n}`"),
    ("csrc/k_frontend_x16.hip", r"void k_msk_frontend_x16\(", "`csrc/k_frontend_x16.hip:{n},"),
    ("csrc/k_frontend_x16.hip", r"void k_msk_frontend_x16_wg4\(", ",:{n},"),
    ("csrc/k_frontend_x16.hip", r"void k_msk_frontend_x16_wg8\(", ",:{n}`"),
    ("csrc/k_offset_search.hip", r"void k_offset_search\(", "`csrc/k_offset_search.hip:{n}`"),
    ("csrc/k_offset_search.hip", r"void k_tie_collect\(", "`:{n},"),
    ("csrc/k_offset_search.hip", r"void k_tie_apply\(", ",:{n}`"),
    ("csrc/opv_offset_host.cpp", r"^void opv_offset_decide_slots\(", "`csrc/opv_offset_host.cpp:{n}`"),
    ("csrc/k_sync_track.hip", r"void k_sync_track\(", "`csrc/k_sync_track.hip:{n}`"),
    ("csrc/k_frame_decode.hip", r"void k_frame_scale\(", "`csrc/k_frame_decode.hip:{n},"),
    ("csrc/k_frame_decode.hip", r"void k_frame_scale_wave\(", ",:{n}` (`k_frame_scale`"),
    ("csrc/k_frame_decode.hip", r"void decode_two\(", "`decode_two` `:{n}`"),
    ("csrc/k_frame_decode.hip", r"uint32_t deint_addr\(", "`csrc/k_frame_decode.hip:{n}` `deint_addr`"),
    ("csrc/k_frame_decode.hip", r"void k_frame_decode\(", "`k_frame_decode` `:{n}`"),
    ("csrc/k_frame_decode.hip", r"LfsrTable kLfsr = ", "`kLfsr` `:{n}`"),
    ("csrc/opv_capi.hip", r"^extern \"C\" int opv_gather_frames\(", "`opv_gather_frames` `csrc/opv_capi.hip:{n}`"),
    ("csrc/opv_capi.hip", r"^extern \"C\" int opv_process\(", "(`csrc/opv_capi.hip:{n}`)"),
    ("csrc/opv_capi.hip", r"^static void tie_host_fn\(", "(`tie_host_fn` `:{n}`"),
    ("csrc/k_tx_modulate.hip", r"void k_tx_encode\(", "`csrc/k_tx_modulate.hip:{n}-"),
    ("csrc/k_tx_modulate.hip", r"void k_tx_modulate\(", "-{n}`"),
    ("csrc/k_channel.hip", r"void k_channel\(", "`csrc/k_channel.hip:{n},"),
    ("csrc/k_channel.hip", r"void k_resample_clock\(", ",{n}`"),
    ("csrc/k_coherent.hip", r"void k_coherent_frontend\(", "`csrc/k_coherent.hip:{n}`"),
]


def test_design_md_anchors_point_at_what_they_name():
    design = (ROOT / "DESIGN.md").read_text()
    assert len(design.encode()) < 31000                       # (the document describes what is shipped; history lives in NOTEBOOK.md)
    for rel, pattern, how in ANCHORS:
        lines = (PKG / rel).read_text().splitlines()
        hits = [i + 1 for i, ln in enumerate(lines) if re.search(pattern, ln)]
        assert len(hits) == 1, (rel, pattern, hits)
        assert how.format(n=hits[0]) in design, f"DESIGN.md: {rel}: /{pattern}/ is at line {hits[0]}, expected the text {how.format(n=hits[0])!r}"


def test_design_md_cites_only_tests_and_files_that_exist():
    design = (ROOT / "DESIGN.md").read_text()
    src = "".join(p.read_text() for p in (ROOT / "tests").glob("test_*.py"))
    names = set(re.findall(r"def (test_\w+)", src))
    for cited in set(re.findall(r"`(test_\w+)", design)) | set(re.findall(r"::(test_\w+)", design)):
        assert (cited in names) if not cited.endswith("_") else any(n.startswith(cited) for n in names), cited
    for rel in set(re.findall(r"`((?:csrc|host|tools)/[\w./]+?)(?::[\d,:\-]+)?`", design)):
        assert (PKG / rel).exists(), rel
    for rel in set(re.findall(r"`(profiles/r\d\d_[\w.]+)`", design)):
        assert (ROOT / rel).exists(), rel

"""The host-code rows of SURVEY.md section 8 (F1 slice segmentation, F2 on-disk formats, the LAZ half of A8) are product
code without kernels; their tests need no GPU and run in the CPU suite.  The round-end driver only runs `-m gpu` on the GPU
box, so the same test functions are collected here a second time under that marker: what the driver records as green then
covers these rows too (same library file, same assertions)."""
import pytest

import test_host_logic as _host
import test_laz as _laz
import test_segmenting as _seg

pytestmark = pytest.mark.gpu

T = _laz.T  # (module-scoped fixture of the LAZ tests)

# F1: treee_segment_slices against oracle/segmenting_restatement.py, the hand-traced two-slice case, the properties
test_f1_library_matches_the_python_restatement = _seg.test_library_matches_the_python_restatement
test_f1_separate_trees_get_one_id_each = _seg.test_separate_trees_get_one_id_each_whatever_the_shuffle
test_f1_hand_traced_two_slices = _seg.test_hand_traced_two_slices
test_f1_voronoi_cells_partition_the_box = _seg.test_voronoi_cells_partition_the_box
test_f1_errors_and_empty_input = _seg.test_errors_and_empty_input
# F2: the hand-assembled bincode vector of InteractiveSave, PLY / JSON text
test_f2_ipc_bincode_layout_and_round_trip = _host.test_ipc_bincode_layout_and_round_trip
test_f2_ply_and_segment_json_exports = _host.test_ply_and_segment_json_exports
test_f2_las_header_chunks_and_errors = _host.test_las_header_chunks_and_errors
# A8 (third-party half): the LAZ decoder against the independent encoder and the committed fixture
test_a8_laz_round_trip = _laz.test_laz_round_trip_through_the_independent_encoder
test_a8_laz_chunk_layouts_and_limits = _laz.test_laz_chunk_layouts_and_limits
test_a8_range_coder_known_answer = _laz.test_range_coder_known_answer
test_a8_laz_golden_fixture = _laz.test_laz_golden_fixture
test_a8_library_encoder_bytes = _laz.test_library_encoder_writes_the_bytes_of_the_independent_one

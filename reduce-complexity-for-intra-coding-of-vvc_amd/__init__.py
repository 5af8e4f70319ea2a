"""MI355X-native VVC intra CU-partition RDO path.  Host-side Python mirror of include/vvcx.h."""
from .vvcx import VvcxEncoder, VvcxError, load_library, TOOL_MRL, TOOL_MIP, TOOL_LFNST, TOOL_MTS, TOOL_JCCR, TOOL_DEPQUANT, TOOL_CU_REUSE, TOOL_CCLM, TOOL_FAST, TOOL_TS, TOOL_RDOQ, TOOL_ISP, TOOL_LMCS, TOOL_WPP, TOOLS_DEFAULT  # noqa: F401
from .synth import synth_frame, slice_params, sao_test_params, alf_test_params, alf_test_frame, ALF_APS_INTS  # noqa: F401
from .vvcx import derive_slice  # noqa: F401
from .sharding import frames_of_rank, units_of_rank, tile_ctus, tile_bounds, timed_steps, gather_ctu_results, gather_payloads, max_over_ranks  # noqa: F401
from .vvcx import distortion_batch, ctx_init, cabac_code_bins, rd_cost_batch, scan_order, transform_quant_batch, PRED_CASE_DTYPE  # noqa: F401
from .forest import forest_from_sklearn, save_forest, load_forest, check_forest  # noqa: F401
from .vvcx import depquant_batch, lfnst_depquant_batch  # noqa: F401

"""rocco_amd -- MI355X-native implementation of ROCCO's per-chromosome solve path.

Re-exports the hot-path functions under the reference's own names (rocco/__init__.py:1-7 star-exports
them from rocco/dp.py and rocco/rocco.py), so `from rocco_amd import solve_chrom_exact` replaces
`from rocco import solve_chrom_exact`.
"""
from .dp import (  # noqa: F401
    build_switch_costs,
    calibrate_selection_penalty,
    objective_value,
    solve_chrom_exact,
    solve_penalized_chain,
)
from .budget import (  # noqa: F401  (rocco/inference.py:988-1148, 1312-1485, 1593-1737)
    estimate_budget_nonnull_fraction_from_empirical_null,
    estimate_budget_nonnull_fraction_from_resampled_null,
    estimate_budget_nonnull_fraction_from_score_track,
    estimate_budget_nonnull_fraction_from_wild_bootstrap_null,
    estimate_empirical_bayes_budgets,
)
from .inference import crossfit_whittaker_baseline  # noqa: F401  (rocco/_baseline.c:16-104)
from .inference import score_centered_wls  # noqa: F401  (rocco/_wls.c)
from .inference import score_loci_wls  # noqa: F401  (rocco/inference.py:302-379)
from .scores import EmpiricalNull, score_peak_counts  # noqa: F401  (rocco/scores.py:120-149, 560-625)
from .readtracks import assemble_chrom_matrix, bigwig_dense_fill  # noqa: F401  (rocco/readtracks.py:141-186, 614-633)
from .rocco import (  # noqa: F401
    chrom_solution_to_bed,
    combine_chrom_results,
    score_central_tendency_chrom,
    solve_cached_chromosomes,
)

__version__ = "0.1.0"

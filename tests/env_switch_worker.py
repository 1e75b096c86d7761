"""Child process of tests/test_gpu_env_switches.py: one mid-size factorisation on the HIP back-end with whatever back-end
environment switches the parent set (they are read once per process), compared with the oracle.  Prints one JSON line."""
import json
import os
import sys

from pangulu_amd import matrices as M

from tests.helpers import factorize, lu_check, max_rel_diff, oracle_library


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "shell"
    mat = M.shell(60, 60) if which == "shell" else M.fem27(24)  # (28 until round 6: 4.5 s per case, 21 cases; the dense paths engage at 24 too)
    # PG_TEST_HIP_OPTIONS="6=0,3=0": back-end options (include/pangulu_platform.h) on top of the test defaults, e.g. the
    # configuration bench.py times (COUNT_FLOPS = 0, PROFILE = 0)
    opts = {int(k): int(v) for k, v in (kv.split("=") for kv in os.environ.get("PG_TEST_HIP_OPTIONS", "").split(",") if kv)}
    gpu = factorize(mat, 256, "hip", hip_options=opts)
    # the oracle's factors of this matrix: from the parent's cache when it keeps one (the sweep runs dozens of settings per matrix)
    cache = os.environ.get("PG_TEST_REF_CACHE")
    if cache and os.path.exists(cache):
        import numpy as np
        import scipy.sparse as sp

        z = np.load(cache)
        n = len(z["L_ptr"]) - 1
        ref = {"L": sp.csc_matrix((z["L_data"], z["L_ind"], z["L_ptr"]), shape=(n, n)), "U": sp.csc_matrix((z["U_data"], z["U_ind"], z["U_ptr"]), shape=(n, n))}
    else:
        ref = factorize(mat, 256, oracle_library("r64"))
        if cache:
            import numpy as np

            L, U = ref["L"].tocsc(), ref["U"].tocsc()
            tmp = cache + ".%d.tmp.npz" % os.getpid()
            np.savez(tmp, L_data=L.data, L_ind=L.indices, L_ptr=L.indptr, U_data=U.data, U_ind=U.indices, U_ptr=U.indptr)
            os.replace(tmp, cache)
    st = gpu["hip_stats"]
    print(json.dumps({
        "dL": max_rel_diff(gpu["L"], ref["L"]), "dU": max_rel_diff(gpu["U"], ref["U"]),
        "residual": gpu["residual"], "lu_check": lu_check(mat, gpu), "factor_check_device": gpu["factor_check"],
        "flop_counted": sum(v["flops"] for v in st.values()), "flop": gpu["info"]["flop"],
        "dense_updates": st["ssssm_dense_mfma"]["tasks"], "dense_solves": st["tstrf"]["dense_path_tasks"],
        "getrf_launches": st["getrf"]["launches"], "chase_launches": st["getrf"]["chase_launches"], "chase_solves": st["tstrf"]["chase_solves"]}))


if __name__ == "__main__":
    main()

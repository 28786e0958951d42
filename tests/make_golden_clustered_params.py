"""Parameters of the complete mid-size measurement on the clustered survey, shared by the generator
(tools/make_golden_clustered.py --full, build container) and tests/test_gpu_clustered.py."""
FULL = dict(n_ref=8e5, n_unk=1.2e6, n_ref_rand=1.6e6, n_unk_rand=1.6e6, rmin=150.0, rmax=1500.0, unit="kpc", rweight=-1.0,
            resolution=30)

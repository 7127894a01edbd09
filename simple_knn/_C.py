from gscream_amd.simple_knn import distCUDA2  # noqa: F401

"""Drop-in for GScream's `simple_knn` extension package (submodules/simple-knn): `from simple_knn._C import distCUDA2`."""

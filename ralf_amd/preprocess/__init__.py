"""Command-line drivers of the offline retrieval build, with the argparse surface of the reference's
image2layout/preprocess/*.py scripts; the scan / selection / embedding work runs on the HIP kernels."""

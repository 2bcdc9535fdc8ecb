"""sha256 over the HIP kernel sources (vit_amd/csrc/*.hip, *.h): ties a committed profile (profiles/pmc_*_traffic.json)
to the kernels it was measured on -- the GPU box's snapshot has no .git, and a commit of the profile itself moves the
head, so the head alone cannot say whether a measurement is stale."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), 'rb').read())
    return h.hexdigest()

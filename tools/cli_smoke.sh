set -e
D=$(mktemp -d)
python - <<PY
import numpy as np
from PIL import Image
import os
d="$D/in"; os.makedirs(d)
rng=np.random.default_rng(0)
for n in ["a b.jpg","České.png","x?y.jpg"]:
    Image.fromarray(rng.integers(0,256,(200,300,3),dtype=np.uint8)).save(os.path.join(d,n))
PY
FCP_WEIGHTS=generated FCP_WEIGHTS_DIR=/nonexistent python -m face_crop_plus_amd -i "$D/in" -o "$D/out" -s 64 -r 256 -st all -dt 0.55 -cn -b 2 2>&1 | tail -2
ls "$D/out" | head; ls "$D" 
FCP_WEIGHTS=generated python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 -m face_crop_plus_amd -i "$D/in" -o "$D/out2" -s 64 -r 256 -st largest -b 2 2>&1 | tail -1
ls "$D/out2" | head

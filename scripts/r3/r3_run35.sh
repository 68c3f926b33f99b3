# GloVe default fits on BA 1M (> 2^31 distinct co-occurrence pairs): the chunked entry ordering
timeout 1500 python scripts/soak.py 1000000 Node2VecGloVeEnsmallen,DeepWalkGloVeEnsmallen > gpurun_out/r3_soak_1m_glove.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_soak_1m_glove.log | tail -12

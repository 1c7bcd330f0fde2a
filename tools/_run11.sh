export AMD_LOG_LEVEL=0
timeout 600 python tools/exp_lockstep.py 4,8,16 sequential,lockstep,staggered:0,staggered:0.25,staggered:0.5,staggered:0.75 2>&1 | tail -5

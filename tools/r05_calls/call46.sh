#!/bin/bash
timeout 300 python tools/two_stream_steps.py --workload lmo_upnp 2>/dev/null | tail -6
timeout 300 python tools/two_stream_steps.py --workload rgb 2>/dev/null | tail -6

#!/bin/bash
echo "== conv_waves (Linears)"; python tools/knob_ab.py conv_waves 8,4 linear 2>&1 | grep " us"
echo "== conv_waves (conv)"; python tools/knob_ab.py conv_waves 8,4 conv 2>&1 | grep " us" | tail -5
echo "== halo_min_m"; python tools/knob_ab.py halo_min_m 2048,256 halo 2>&1 | grep " us" | tail -2
echo "== halo_brick"; python tools/knob_ab.py halo_brick 0,1,2 halo 2>&1 | grep " us" | head -5
echo "== halo_min_cout"; python tools/knob_ab.py halo_min_cout 16,64 halo 2>&1 | grep " us" | sed -n 3p

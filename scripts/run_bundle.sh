#!/bin/bash
# scratch: the command bundle of the latest gpurun call
cd /root/repo; mkdir -p gpurun_out

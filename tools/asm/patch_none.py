import sys, shutil
shutil.copy(sys.argv[1], sys.argv[2])

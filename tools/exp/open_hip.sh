D=/tmp/zwz_openhip_$$; mkdir -p $D; python3 -c "
import os
for i in range(8000): open('$D/f%05d.bin' % i, 'wb').write(b'x' * 262144)
"
tools/exp/open_hip.bin $D 0; tools/exp/open_hip.bin $D 1; rm -rf $D

/* c-ray-hip build: stand-in for the file the reference's CMake generates from src/utils/gitsha1.c.in (CMakeLists.txt:69-71).
 * The hash is what the cluster handshake compares (src/utils/protocol/worker.c:61-70: a worker refuses a master built from another
 * commit), so it has to be the hash of the reference tree this program was built from — the renderer backend is not part of it.
 * /root/reference carries no git metadata: its builds here (oracle/_ref) say "NoHash", and so does this one. CRAY_HIP_GITHASH
 * overrides it for a cluster whose masters were built from a real checkout. */
#include <stdlib.h>
char *gitHash(void) {
	char *env = getenv("CRAY_HIP_GITHASH");
	return (env && *env) ? env : "NoHash";
}
